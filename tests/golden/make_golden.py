"""Generates tests/golden/*.npz from oracle/torch_ref.py (the restatement built from the ATen ops the
reference calls; float64 autograd so the vectors are independent of fp32 summation order).

The reference itself cannot produce vectors: it holds no tests or fixtures and does not build in this
image (SURVEY.md sections 4 and 8c).  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from oracle import torch_ref as T  # noqa: E402


def one_case(stage, with_gt, occupancy, seed, n_rays=24):
    sc = scenes.make_scene(seed, scenes.SMALL_GRID_SHAPES, grid_std=0.3, bias_std=0.1)
    rays = scenes.make_rays(seed + 100, n_rays, sc["bound"], n_frames=2, zero_frac=0.15)
    rng = np.random.default_rng(seed + 200)
    N = rays["rays_o"].shape[0]
    g_rgb, g_d, g_v = rng.standard_normal((N, 3)), rng.standard_normal(N), rng.standard_normal(N)
    f64 = torch.float64
    grids = {k: torch.tensor(v, dtype=f64)[None].requires_grad_(True) for k, v in sc["grids"].items()}
    decs = {k: torch.tensor(v, dtype=f64).requires_grad_(True) for k, v in sc["decoders"].items()}
    ro = torch.tensor(rays["rays_o"], dtype=f64).requires_grad_(True)
    rd = torch.tensor(rays["rays_d"], dtype=f64).requires_grad_(True)
    gd = torch.tensor(rays["gt_depth"], dtype=f64) if with_gt else None
    rgb, d, var, w, z, raw = T.render_batch_ray(grids, decs, rd, ro, stage, gd, torch.tensor(sc["bound"], dtype=f64),
                                                occupancy=occupancy, return_aux=True)
    L = (rgb * torch.tensor(g_rgb)).sum() + (d * torch.tensor(g_d)).sum() + (var * torch.tensor(g_v)).sum()
    L.backward()
    out = dict(stage=stage, with_gt=with_gt, occupancy=occupancy, seed=seed, bound=sc["bound"],
               rays_o=rays["rays_o"], rays_d=rays["rays_d"], gt_depth=rays["gt_depth"], gt_color=rays["gt_color"],
               g_rgb=g_rgb, g_depth=g_d, g_var=g_v,
               rgb=rgb.detach().numpy(), depth=d.detach().numpy(), var=var.detach().numpy(),
               weights=w.detach().numpy(), z=z.detach().numpy(), raw=raw.detach().numpy(),
               g_rays_o=ro.grad.numpy(), g_rays_d=rd.grad.numpy())
    for k in scenes.LEVELS:
        out["grid_" + k] = sc["grids"][k]
        out["dec_" + k] = sc["decoders"][k]
        if grids[k].grad is not None:
            out["g_grid_" + k] = grids[k].grad[0].numpy().astype(np.float32)
        if decs[k].grad is not None:
            out["g_dec_" + k] = decs[k].grad.numpy()
    return out


def main():
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(0)
    cases = [("coarse", True, False, 11), ("middle", True, False, 12), ("fine", True, False, 13),
             ("color", True, False, 14), ("color", False, False, 15), ("color", True, True, 16),
             ("middle", False, True, 17)]
    for stage, gt, occ, seed in cases:
        out = one_case(stage, gt, occ, seed)
        name = "render_%s_%s_%s.npz" % (stage, "gt" if gt else "nogt", "occ" if occ else "dens")
        np.savez_compressed(os.path.join(HERE, name), **out)
        print("wrote", name)


if __name__ == "__main__":
    main()
