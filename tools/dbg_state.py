"""debug: where do optimised grids differ from the oracle after mapping steps?  python tools/dbg_state.py [steps] [sort_mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
from oracle.nso import Oracle
from gpu_util import cu, make_ctx
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sort = int(sys.argv[2]) if len(sys.argv) > 2 else -1
o = Oracle("f32"); o.lib.nso_set_num_threads(16)
o64 = Oracle("f64"); o64.lib.nso_set_num_threads(16)
sc = scenes.make_scene(51)
rays = scenes.make_rays(52, 1000, sc["bound"], n_frames=5)
ctx = make_ctx(sc, trainable=["color"]); ctx.set_sort_mode(sort)
ro, rd, gd, gc = cu(rays["rays_o"]), cu(rays["rays_d"]), cu(rays["gt_depth"]), cu(rays["gt_color"])
loss = torch.zeros(1, device="cuda")
# gradients of one step
ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss)
gg = {k: ctx.grid_download(k, grad=True) for k in ("middle", "fine", "color")}
op = o.opts(sc["bound"])
def oracle_grads(o):
    op = o.opts(sc["bound"])
    fw = o.render_forward(op, sc["grids"], sc["decoders"], "color", rays["rays_o"], rays["rays_d"], rays["gt_depth"])
    l, g_d, g_c = o.loss_map(fw["depth"], fw["rgb"], rays["gt_depth"], rays["gt_color"], 0.5, True)
    return o.render_backward(op, sc["grids"], sc["decoders"], "color", rays["rays_o"], rays["rays_d"], rays["gt_depth"], -1.0, g_c, g_d, None, want_rays=False), g_d, g_c, fw
bw, g_d, g_c, fw = oracle_grads(o)
bw64, g_d64, g_c64, fw64 = oracle_grads(o64)
print("seed gradient sign differences f32 vs f64: depth", int((np.sign(g_d) != np.sign(g_d64)).sum()), "colour", int((np.sign(g_c) != np.sign(g_c64)).sum()))
for k in gg:
    a, r, r64 = gg[k], bw["g_grids"][k], bw64["g_grids"][k]
    print(k, "grad rel_l2 hip-vs-f32 %.2e hip-vs-f64 %.2e f32-vs-f64 %.2e" % (scenes.rel_l2(a, r), scenes.rel_l2(a, r64), scenes.rel_l2(r, r64)),
          "nonzero hip %d oracle %d; sign mismatches %d; hip zero where oracle nonzero %d (max |g| there %.2e); reverse %d" % (
          (a != 0).sum(), (r != 0).sum(), ((np.sign(a) != np.sign(r)) & (a != 0) & (r != 0)).sum(), ((a == 0) & (r != 0)).sum(),
          np.abs(r[(a == 0) & (r != 0)]).max() if ((a == 0) & (r != 0)).any() else 0.0, ((a != 0) & (r == 0)).sum()))
    d = np.abs(a - r)
    idx = np.argsort(d.ravel())[::-1][:5]
    for i in idx:
        u = np.unravel_index(i, a.shape)
        print("   ", u, "hip %.6e f32 %.6e f64 %.6e" % (a[u], r[u], r64[u]))
# one Adam step
lr = [0.005, 0.0, 0.005, 0.005, 0.005, 0.0]
ctx.adam_step(lr); ctx.sync()
for prec, oo, b in (("f32", o, bw), ("f64", o64, bw64)):
    for k in ("middle", "fine", "color"):
        p = sc["grids"][k].astype(oo.dt).copy(); m = np.zeros_like(p); v = np.zeros_like(p)
        oo.adam_step(p, b["g_grids"][k], m, v, 0.005, 1)
        got = ctx.grid_download(k)
        d = np.abs(got - p)
        g = np.abs(b["g_grids"][k])
        print(prec, k, "state rel_l2 %.2e; elements off by > 1e-5: %d; their |g| range %.1e..%.1e; frac of |g| in (0,1e-6): %.4f" % (
            scenes.rel_l2(got, p), (d > 1e-5).sum(), g[d > 1e-5].min() if (d > 1e-5).any() else 0, g[d > 1e-5].max() if (d > 1e-5).any() else 0,
            ((g > 0) & (g < 1e-6)).mean()))
        big = g >= 1e-5
        print("      restricted to |g| >= 1e-5 (%d elements): rel_l2 of the update %.2e" % (big.sum(), scenes.rel_l2((got - sc["grids"][k])[big], (p - sc["grids"][k])[big])))
