"""N mapping steps (colour stage) for rocprofv3 runs: python tools/prof_step.py [rays] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sc = scenes.make_scene(42)
r = scenes.make_rays(1234, N, sc["bound"], n_frames=5)
ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"]); ctx.decoder_set_trainable("color", True)
cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
ro, rd, gd, gc = cu(r["rays_o"]), cu(r["rays_d"]), cu(r["gt_depth"]), cu(r["gt_color"])
loss = torch.zeros(1, device="cuda")
for i in range(K):
    ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss)
    ctx.adam_step([0.005, 0.0, 0.005, 0.005, 0.005, 0.0])
ctx.sync()
print("ok", float(loss))
