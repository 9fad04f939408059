"""experiment: per-wave phase timestamps of the trainable-decoder backward role (libnsk_exp.so)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NSK_LIB"] = os.path.join(ROOT, "nice-slam-cpp_amd", "csrc", "libnsk_exp.so")
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
sc = scenes.make_scene(42)
r = scenes.make_rays(1234, 1000, sc["bound"], n_frames=5)
ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"]); ctx.decoder_set_trainable("color", True)
cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
ro, rd, gd, gc = cu(r["rays_o"]), cu(r["rays_d"]), cu(r["gt_depth"]), cu(r["gt_color"])
L = pkg.nsk.lib()
for i in range(3):
    ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3); ctx.zero_grads()
ctx.sync()
L.nsk_dbg_enable(ctx.h)
ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3)
buf = np.zeros(1024 * 16 * 16, np.uint64)
L.nsk_dbg_read(ctx.h, buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(1024, 16, 16).astype(np.int64)
names = ["setup", "OUT", "l4", "l3", "l2", "l1", "l0", "dB+rays", "scatter"]
for wg in (0, 1, 50, 150):
    for wv in (0, 5):
        x = t[wg, wv]
        if x[9] == 0: continue
        print("wg %3d wave %d: " % (wg, wv) + " ".join("%s %d" % (names[k], x[k + 1] - x[k]) for k in range(9)) + " | total %d" % (x[9] - x[0]))
