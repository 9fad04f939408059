"""experiment: per-wave phase timestamps of the forward multi kernel (libnsk_exp.so)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NSK_LIB"] = os.path.join(ROOT, "nice-slam-cpp_amd", "csrc", "libnsk_exp.so")
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
sc = scenes.make_scene(42)
r = scenes.make_rays(1234, 1000, sc["bound"], n_frames=5)
ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"])
cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
ro, rd, gd = cu(r["rays_o"]), cu(r["rays_d"]), cu(r["gt_depth"])
L = pkg.nsk.lib()
for i in range(3):
    ctx.render_forward("color", ro, rd, gd)
ctx.sync()
L.nsk_dbg_enable(ctx.h)
ctx.render_forward("color", ro, rd, gd)
buf = np.zeros(1024 * 16 * 16, np.uint64)
L.nsk_dbg_read(ctx.h, buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(1024, 16, 16).astype(np.int64)
used = t[:, :, 0] > 0
t0 = t[:, :, 0][used].min()
for wg in (0, 1, 100, 200, 255):
    for wv in (0, 4):
        x = t[wg, wv]
        if x[0] == 0: continue
        rel = [(int(v) - int(t0)) if v > 0 else -1 for v in x[:14]]
        print("wg %3d wave %d: start %6d staged %6d | " % (wg, wv, rel[0], rel[1]) + " ".join("[prep %d emb %d chain %d]" % (rel[2+3*k], rel[3+3*k]-rel[2+3*k], rel[4+3*k]-rel[3+3*k]) for k in range(4) if rel[4+3*k] > 0))
ends = t[:, :, 2:14].max(axis=2)[used]
print("kernel span (cycles @100MHz memtime?):", ends.max() - t0, "first-start spread", t[:, :, 0][used].max() - t0)
