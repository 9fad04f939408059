"""experiment: time map_step kernels under debug switches (needs libnsk_exp.so built with -DNSK_EXPERIMENT)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NSK_LIB"] = os.path.join(ROOT, "nice-slam-cpp_amd", "csrc", "libnsk_exp.so")
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sc = scenes.make_scene(42)
r = scenes.make_rays(1234, N, sc["bound"], n_frames=5)
ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"]); ctx.decoder_set_trainable("color", True)
cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
ro, rd, gd, gc = cu(r["rays_o"]), cu(r["rays_d"]), cu(r["gt_depth"]), cu(r["gt_color"])
loss = torch.zeros(1, device="cuda")
# NOT offered: dropping the phase barriers (bit 12).  The scatter's run table lives in the panel's head, so without barriers another wave's
# panel writes turn into scatter addresses: the run faulted the GPU (round 2).  Every ablation here keeps the barriers.
for name, extra in [("base", 0), ("no_scatter", 1 << 9), ("no_tiles", 1 << 13), ("no_tiles_put", (1 << 13) | (1 << 14)), ("no_tiles_put_scatter", (1 << 9) | (1 << 13) | (1 << 14))]:
    with torch.cuda.stream(ctx.tstream):
        for i in range(3):
            ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3 | extra, loss=loss); ctx.zero_grads()
        ctx.profile_begin()
        for i in range(10):
            ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3 | extra, loss=loss); ctx.zero_grads()
        p = ctx.profile_end()
    print(name, {k: round(1e3 * ms / c, 1) for k, (c, ms) in p.items() if "bwd" in k})
