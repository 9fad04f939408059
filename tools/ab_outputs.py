"""Bit-level A/B of two builds of libnsk.so (NSK_LIB selects the library; run once per build, compare the printed digests): the K3 colour-stage
forward at full size (rendered colour / depth / variance / weights, the ReLU bits and ReLU inputs of all three decoders) and, in the deterministic
mode (fixed summation order), the gradients of a 300-ray step.  A change that claims "the same bits" (an instruction-selection change such as the
fma_mix operand split) must leave every digest unchanged.   usage: NSK_LIB=<lib> python tools/ab_outputs.py"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
from gpu_util import cu, make_ctx
import test_gpu_configs as tc

dg = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
sc, rays, stage, _ = tc._strict_case("K3-color")
gmax = float(rays["gt_depth"].max())
N = rays["rays_o"].shape[0]; M = N * 48
ro, rd, gd, gc = cu(rays["rays_o"]), cu(rays["rays_d"]), cu(rays["gt_depth"]), cu(rays["gt_color"])
for mode in (2, 1, 0):
    ctx = make_ctx(sc, trainable=["color"])
    ctx.set_matmul_mode(mode)
    rgb, depth, var, w = ctx.render_forward("color", ro, rd, gd, gmax)
    loss = torch.zeros(1, device="cuda")
    ctx.map_step("color", ro, rd, gd, gc, gmax, 0.5, True, flags=3, loss=loss)
    out = ["fwd mode %d:" % mode, "rgb", dg(rgb.cpu().numpy()), "depth", dg(depth.cpu().numpy()), "w", dg(w.cpu().numpy()), "loss", repr(float(loss))]
    for k in ("middle", "fine", "color"):
        out += [k, dg(ctx.debug_relu_bits(k, M)), dg(ctx.debug_preact(k, ro, rd, M))]
    print(" ".join(out))
    ctx.close()
n = 300
for bm in (2, 0):
    ctx = make_ctx(sc, trainable=["color"])
    ctx.set_tuning("deterministic", 1)
    ctx.set_backward_mode(bm)
    loss = torch.zeros(1, device="cuda")
    ctx.map_step("color", ro[:n].contiguous(), rd[:n].contiguous(), gd[:n].contiguous(), gc[:n].contiguous(), gmax, 0.5, True, flags=3, loss=loss)
    print("deterministic step, backward mode %d:" % bm, " ".join("%s %s" % (k, dg(ctx.grid_download(k, grad=True))) for k in ("middle", "fine", "color")),
          "decoder", dg(ctx.decoder_download("color", grad=True)))
    ctx.close()
