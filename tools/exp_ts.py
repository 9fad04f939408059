"""experiment: per-workgroup start/end stamps of the fwd / bwd multi kernels (libnsk_exp.so, `make -C nice-slam-cpp_amd/csrc exp`)
python tools/exp_ts.py [rays] [sort_mode] [frozen_cost ...]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NSK_LIB"] = os.environ.get("NSK_EXP_LIB", os.path.join(ROOT, "nice-slam-cpp_amd", "csrc", "libnsk_exp.so"))
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
SORT = int(sys.argv[2]) if len(sys.argv) > 2 else -1
COSTS = [int(a) for a in sys.argv[3:]] or [0]
sc = scenes.make_scene(42)
r = scenes.make_rays(1234, N, sc["bound"], n_frames=5)
ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"]); ctx.decoder_set_trainable("color", True)
ctx.set_sort_mode(SORT)
cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
ro, rd, gd, gc = cu(r["rays_o"]), cu(r["rays_d"]), cu(r["gt_depth"]), cu(r["gt_color"])
loss = torch.zeros(1, device="cuda")
lib = C.CDLL(os.environ["NSK_LIB"])
def report(tag, flags=3):
    for i in range(5):
        ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=flags, loss=loss); ctx.zero_grads()
    ctx.sync()
    ts = np.zeros((2, 1024, 4), np.uint64)
    lib.nsk_dbg_read_ts(ctx.h, ts.ctypes.data_as(C.c_void_p))
    out = [tag]
    for k, name in ((0, "fwd"), (1, "bwd")):
        t = ts[k]; used = t[:, 1] > 0
        t = t[used].astype(np.int64)
        if t.shape[0] == 0:
            continue
        t0 = t[:, 0].min()
        out.append("%s span %.1f" % (name, (t[:, 1].max() - t0) / 100.0))
        for role in np.unique(t[:, 2]):
            m = t[:, 2] == role
            e = (t[m, 1] - t0) / 100.0
            out.append("r%d[%d wg] end med %.1f max %.1f" % (role, m.sum(), np.median(e), e.max()))
    print(" | ".join(out), flush=True)

for cost in COSTS:
    ctx.set_tuning("frozen_cost", cost)
    report("sort %d cost %d base" % (SORT, cost))
    for bits, name in ((1, "no_atomics(walk kept)"), (2, "no_scatter_mfma"), (4, "no_voxel_prefetch"), (7, "scatter walk only")):
        lib.nsk_dbg_set(ctx.h, bits); report("sort %d cost %d %s" % (SORT, cost, name)); lib.nsk_dbg_set(ctx.h, 0)
    report("sort %d cost %d no_scatter" % (SORT, cost), 3 | (1 << 9))
    # (never drop the barriers, bit 12: the scatter's run table shares LDS with the panel -- see tools/exp_bwd.py)
    report("sort %d cost %d no_tiles_put" % (SORT, cost), 3 | (1 << 13) | (1 << 14))
    report("sort %d cost %d no_tiles_put_scatter" % (SORT, cost), 3 | (1 << 9) | (1 << 13) | (1 << 14))
