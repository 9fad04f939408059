"""experiment: s_memtime stamps inside the merged-phase trainable body (decode_bwd_train_m_body, libnsk_exp.so) on a bench workload, with
parts switched off (never the barriers).   python tools/exp_ph3.py [K3|K2|K4] [rays]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NSK_LIB"] = os.path.join(ROOT, "nice-slam-cpp_amd", "csrc", "libnsk_exp.so")
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes, bench
WL = sys.argv[1] if len(sys.argv) > 1 else "K3"
wl = bench.workloads()[WL]
N = int(sys.argv[2]) if len(sys.argv) > 2 else wl["rays"]
sc = scenes.make_scene(42, scenes.grid_shapes_for(wl["bound"]), bound=wl["bound"])
r = scenes.make_rays(1234, N, sc["bound"], n_frames=5, cam_seed=4242, up=wl["up"], **wl["cam"])
ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"]); ctx.decoder_set_trainable("color", True)
cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
ro, rd, gd, gc = cu(r["rays_o"]), cu(r["rays_d"]), cu(r["gt_depth"]), cu(r["gt_color"])
loss = torch.zeros(1, device="cuda")
lib = C.CDLL(os.environ["NSK_LIB"])
names = {0: "top", 1: "E/XC puts", 2: "OUT", 3: "L4", 4: "L3", 5: "L2", 6: "L1", 7: "L0", 8: "gemm_e+stage_a", 9: "cos+DB", 10: "stage_b", 11: "scatter"}

def report(tag, flags=3):
    with torch.cuda.stream(ctx.tstream):
        for i in range(4):
            ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=flags, loss=loss); ctx.zero_grads()
    ctx.sync()
    ph = np.zeros((8, 8, 96), np.uint64)
    lib.nsk_dbg_read_ph(ctx.h, ph.ctypes.data_as(C.c_void_p))
    ph = ph.astype(np.int64)
    med = lambda a: int(np.median(a))
    print("== %s (%s, %d rays) ==" % (tag, WL, N))
    print("  prologue: image copy %d, first stage %d;  epilogue: flush %d;  whole body %d cycles" %
          (med(ph[:, :, 29] - ph[:, :, 28]), med(ph[:, :, 30] - ph[:, :, 29]), med(ph[:, :, 13] - ph[:, :, 12]), med(ph[:, :, 13] - ph[:, :, 28])))
    for itn, base in (("first", 32), ("second", 64), ("last", 0)):
        t = ph[:, :, base:base + 12]
        d = np.diff(t, axis=2)
        tot = t[:, :, 11] - t[:, :, 0]
        print("  %-6s iteration %6d cycles: " % (itn, med(tot)) + " ".join("%s %d" % (names[k + 1], med(d[:, :, k])) for k in range(11)))
    for itn, base in (("second", 64), ("last", 0)):
        print("  %-6s iteration, after layer 0: e-image stores %d, barrier %d, e-part products %d, stage_a issue %d" %
              (itn, med(ph[:, :, base + 14] - ph[:, :, base + 7]), med(ph[:, :, base + 15] - ph[:, :, base + 14]), med(ph[:, :, base + 16] - ph[:, :, base + 15]),
               med(ph[:, :, base + 8] - ph[:, :, base + 16])))
    t = ph[:, :, 20:27]
    print("  layer 2 (last iteration): mask+FT %d, puts %d, barrier %d, tiles %d, barrier %d, WT %d   (per wave of wg 0, tiles: %s)" %
          tuple([med(t[:, :, k + 1] - t[:, :, k]) for k in range(6)] + [" ".join(str(int(x)) for x in (ph[0, :, 24] - ph[0, :, 23]))]))
    sys.stdout.flush()

report("base")
report("stage_a without the sample loads (z, ray)", 3 | (1 << 10))
report("stage_a without g_raw and the ReLU bits", 3 | (1 << 11))
report("stage_a without h4", 3 | (1 << 15))
report("stage_a without any load but perm", 3 | (1 << 10) | (1 << 11) | (1 << 15))
report("no scatter", 3 | (1 << 9))
report("no tiles", 3 | (1 << 13))
report("no puts", 3 | (1 << 14))
report("no tiles, no puts", 3 | (1 << 13) | (1 << 14))
report("no tiles, no puts, no scatter", 3 | (1 << 9) | (1 << 13) | (1 << 14))
