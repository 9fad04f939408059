"""experiment (round 4): per-role end stamps of the backward of a BUNDLE-ADJUSTMENT iteration (flags = grids | decoders | rays: k_decode_bwd_multi<true>,
ray order) against a plain mapping iteration of the same batch (libnsk_exp.so).  python tools/exp_ts_ba.py [rays]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NSK_LIB"] = os.path.join(ROOT, "nice-slam-cpp_amd", "csrc", "libnsk_exp.so")
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
COSTS = [int(a) for a in sys.argv[2:]] or [0]
sc = scenes.make_scene(42)
r = scenes.make_rays(1234, N, sc["bound"], n_frames=5)
ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"]); ctx.decoder_set_trainable("color", True)
cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
ro, rd, gd, gc = cu(r["rays_o"]), cu(r["rays_d"]), cu(r["gt_depth"]), cu(r["gt_color"])
g = (torch.zeros_like(ro), torch.zeros_like(rd))
loss = torch.zeros(1, device="cuda")
lib = C.CDLL(os.environ["NSK_LIB"])
for cost in COSTS:
  ctx.set_tuning("frozen_cost_rays", cost)
  for tag, flags, sort in (("mapping step (sorted)", 3, -1), ("BA step (ray order)", 7, 0), ("BA step (sorted), frozen_cost_rays %d" % cost, 7, 1)):
    ctx.set_sort_mode(sort)
    for i in range(5):
      ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=flags, loss=loss, **({"g_rays": g} if flags & 4 else {})); ctx.zero_grads()
    ctx.sync()
    ts = np.zeros((2, 1024, 4), np.uint64)
    lib.nsk_dbg_read_ts(ctx.h, ts.ctypes.data_as(C.c_void_p))
    out = [tag]
    t = ts[1]; t = t[t[:, 1] > 0].astype(np.int64)
    t0 = t[:, 0].min()
    out.append("bwd span %.1f us" % ((t[:, 1].max() - t0) / 100.0))
    for role in np.unique(t[:, 2]):
        m = t[:, 2] == role
        e = (t[m, 1] - t0) / 100.0
        out.append("role %d [%d wg] end med %.1f max %.1f" % (role, m.sum(), np.median(e), e.max()))
    print(" | ".join(out), flush=True)
