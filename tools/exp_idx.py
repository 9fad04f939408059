"""experiment: range audit of every per-sample index of the decoder bodies (NSK_IDX, libnsk_exp.so): mapping steps at the bench workloads
(cell-sorted and ray order, ragged batch sizes) and a tracking step; prints the out-of-range counts by site (all must be 0)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NSK_LIB"] = os.path.join(ROOT, "nice-slam-cpp_amd", "csrc", "libnsk_exp.so")
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes, bench
lib = C.CDLL(os.environ["NSK_LIB"])
cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
total = np.zeros(8, np.int64)
for WL, rays in (("K3", 5000), ("K3", 4999), ("K2", 1000), ("K2", 333), ("K4", 1250), ("K2", 7)):
    wl = bench.workloads()[WL]
    sc = scenes.make_scene(42, scenes.grid_shapes_for(wl["bound"]), bound=wl["bound"])
    r = scenes.make_rays(1234, rays, sc["bound"], n_frames=5, cam_seed=4242, up=wl["up"], **wl["cam"])
    ro, rd, gd, gc = cu(r["rays_o"]), cu(r["rays_d"]), cu(r["gt_depth"]), cu(r["gt_color"])
    for sort in (1, 0):
        ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"]); ctx.decoder_set_trainable("color", True)
        ctx.set_sort_mode(sort)
        loss = torch.zeros(1, device="cuda")
        with torch.cuda.stream(ctx.tstream):
            for stage, flags in (("color", 3), ("fine", 1), ("color", 7)):
                g = (torch.empty_like(ro), torch.empty_like(rd)) if flags & 4 else None
                ctx.map_step(stage, ro, rd, gd, gc, -1.0, 0.5, stage == "color", flags=flags, loss=loss, **({"g_rays": g} if g else {}))
                ctx.adam_step([0.005, 0.0, 0.005, 0.005, 0.005, 0.0])
            g = (torch.empty_like(ro), torch.empty_like(rd))
            ctx.track_step("color", ro, rd, gd, gc, -1.0, 0.5, True, True, True, flags=4, loss=loss, g_rays=g)
        ctx.sync()
        oob = np.zeros(8, np.uint32)
        lib.nsk_dbg_read_oob(ctx.h, oob.ctypes.data_as(C.c_void_p), 1)
        print(WL, rays, "rays, sort", sort, "-> out-of-range indices by site [perm entry, sample, slot, tile, ray]:", oob[:5].tolist(), flush=True)
        total += oob
        ctx.close()
print("TOTAL", total[:5].tolist())
assert not total.any()
