"""experiment: where the cell sort starts to pay -- ms per colour-stage mapping step (reference grids) in ray order and cell-sorted order"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
import bench
sc = scenes.make_scene(42)
cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
lr = bench.STAGE_LR["color"]
for n in (64, 128, 200, 300, 500, 1000):
    r = scenes.make_rays(1234, n, sc["bound"], n_frames=5)
    t = [cu(r[k]) for k in ("rays_o", "rays_d", "gt_depth", "gt_color")]
    out = []
    for mode in (0, 1):
        ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"]); ctx.decoder_set_trainable("color", True)
        ctx.set_sort_mode(mode)
        loss = torch.zeros(1, device="cuda")
        with torch.cuda.stream(ctx.tstream):
            for _ in range(20):
                ctx.map_step("color", *t, -1.0, 0.2, True, flags=3, loss=loss); ctx.adam_step(lr)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(300):
                ctx.map_step("color", *t, -1.0, 0.2, True, flags=3, loss=loss); ctx.adam_step(lr)
            torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 300 * 1e6)
        ctx.close()
    print("%5d rays (%6d samples): ray order %.1f us, cell-sorted %.1f us per step" % (n, n * 48, out[0], out[1]), flush=True)
