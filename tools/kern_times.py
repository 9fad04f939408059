"""per-kernel HIP-event times of the colour-stage mapping step: python tools/kern_times.py [rays] ; NSK_LIB selects the build"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sc = scenes.make_scene(42)
r = scenes.make_rays(1234, N, sc["bound"], n_frames=5)
cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
ro, rd, gd, gc = cu(r["rays_o"]), cu(r["rays_d"]), cu(r["gt_depth"]), cu(r["gt_color"])
for mode in (0, 1):
    ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"]); ctx.decoder_set_trainable("color", True)
    ctx.set_matmul_mode(mode)
    loss = torch.zeros(1, device="cuda")
    lr = [0.005, 0.0, 0.005, 0.005, 0.005, 0.0]
    with torch.cuda.stream(ctx.tstream):
        for i in range(5):
            ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss); ctx.adam_step(lr)
        ctx.profile_begin()
        for i in range(20):
            ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss); ctx.adam_step(lr)
        p = ctx.profile_end()
    d = {k: round(1e3 * ms / c, 1) for k, (c, ms) in p.items()}
    print(os.path.basename(os.environ.get("NSK_LIB", "libnsk.so")), "mode", mode, "total", round(sum(d.values()), 1), d, "loss %.6f" % float(loss))
    del ctx
