"""debug: K4 full batch, one step: which elements of the optimised grids differ from the oracle's and what were their gradients"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
from oracle.nso import Oracle
from gpu_util import cu, make_ctx
o = Oracle("f32"); o.lib.nso_set_num_threads(16)
sc = scenes.make_scene(73, scenes.grid_shapes_for(scenes.K4_BOUND), bound=scenes.K4_BOUND)
rays = scenes.make_rays(74, 10000, sc["bound"], n_frames=5, **scenes.CAM_NICE_SLAM)
ctx = make_ctx(sc, trainable=["color"])
ro, rd, gd, gc = cu(rays["rays_o"]), cu(rays["rays_d"]), cu(rays["gt_depth"]), cu(rays["gt_color"])
loss = torch.zeros(1, device="cuda")
ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss)
gg = {k: ctx.grid_download(k, grad=True) for k in ("middle", "fine", "color")}
ctx.adam_step([0.005, 0.0, 0.005, 0.005, 0.005, 0.0]); ctx.sync()
op = o.opts(sc["bound"])
fw = o.render_forward(op, sc["grids"], sc["decoders"], "color", rays["rays_o"], rays["rays_d"], rays["gt_depth"])
l, g_d, g_c = o.loss_map(fw["depth"], fw["rgb"], rays["gt_depth"], rays["gt_color"], 0.5, True)
bw = o.render_backward(op, sc["grids"], sc["decoders"], "color", rays["rays_o"], rays["rays_d"], rays["gt_depth"], -1.0, g_c, g_d, None, want_rays=False)
for k in gg:
    p = sc["grids"][k].copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    o.adam_step(p, bw["g_grids"][k], m, v, 0.005, 1)
    got = ctx.grid_download(k)
    d = np.abs(got - p)
    print(k, "state rel_l2 %.2e grad rel_l2 %.2e; off by > 1e-4: %d" % (scenes.rel_l2(got, p), scenes.rel_l2(gg[k], bw["g_grids"][k]), (d > 1e-4).sum()))
    idx = np.argsort(d.ravel())[::-1][:12]
    for i in idx:
        u = np.unravel_index(i, d.shape)
        print("   ", tuple(int(x) for x in u), "init %.5f hip %.5f oracle %.5f | g hip %.4e oracle %.4e" % (sc["grids"][k][u], got[u], p[u], gg[k][u], bw["g_grids"][k][u]))
    big = d > 1e-4
    if big.any():
        zs = np.unique(np.argwhere(big)[:, 1]); ys = np.unique(np.argwhere(big)[:, 2]); xs = np.unique(np.argwhere(big)[:, 3])
        print("    z", zs[:20], "y", ys[:20], "x", xs[:20], "shape", d.shape)
