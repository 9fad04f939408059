"""experiment: s_memtime stamps inside k_sample's workgroup 300 at 5000 rays (libnsk_exp.so): python tools/exp_sample.py"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NSK_LIB"] = os.path.join(ROOT, "nice-slam-cpp_amd", "csrc", "libnsk_exp.so")
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
import bench
wl = bench.workloads()["K3"]
sc = scenes.make_scene(42, scenes.grid_shapes_for(wl["bound"]), bound=wl["bound"])
r = scenes.make_rays(1234, 5000, sc["bound"], n_frames=5, cam_seed=4242, up=wl["up"], **wl["cam"])
ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"]); ctx.decoder_set_trainable("color", True)
cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
ro, rd, gd, gc = cu(r["rays_o"]), cu(r["rays_d"]), cu(r["gt_depth"]), cu(r["gt_color"])
gm = float(r["gt_depth"].max())
loss = torch.zeros(1, device="cuda")
lib = C.CDLL(os.environ["NSK_LIB"])
names = ["ray loads..z written", "table init", "cell keys", "LDS table (CAS, counts)", "barrier", "global histogram adds", "skey / srank stores"]
for rep in range(3):
    for i in range(3):
        ctx.map_step("color", ro, rd, gd, gc, gm, 0.5, True, flags=3, loss=loss); ctx.zero_grads()
    ctx.sync()
    ts = np.zeros((2, 1024, 4), np.uint64)
    lib.nsk_dbg_read_ts(ctx.h, ts.ctypes.data_as(C.c_void_p))
    t = ts[0, 1000:1008, 0].astype(np.int64)
    d = np.diff(t)
    print("k_sample workgroup 300, cycles: " + ", ".join("%s %d" % (n, x) for n, x in zip(names, d)) + "; total %d" % (t[-1] - t[0]))
