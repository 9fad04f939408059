"""experiment: s_memtime stamps inside the trainable-decoder backward (libnsk_exp.so): cycles between points of the last iteration"""
import os, sys, ctypes as C
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
lib = C.CDLL(os.environ["NSK_LIB"])
for i in range(5):
    ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss); ctx.zero_grads()
ctx.sync()
ph = np.zeros((8, 8, 96), np.uint64)
lib.nsk_dbg_read_ph(ctx.h, ph.ctypes.data_as(C.c_void_p))
ph = ph.astype(np.int64)
names_unused = ["top", "fwd done", "OUT done", "L4", "sc0", "L3", "L2", "L1", "L0", "sc4 done", "DB/rays done", "staged", "tail flush", "slab flush"]
pts = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
for b in (0, 3):
    for w in (0, 5):
        t = ph[b, w]
        print("wg", b, "wave", w, " ".join("%s:%d" % (k, t[k] - t[0]) for k in pts))
d = np.diff(ph[:, :, :13], axis=2)
t = np.median(ph.reshape(-1, 32), axis=0)
print("layer 3: FT+FC phase %d, W3(e) phase %d, W3H phase %d, gemm_e %d, rest %d" % (t[13] - t[3], t[14] - t[13], t[15] - t[14], t[16] - t[15], t[4] - t[16]))
print("stage+reload %d, scatter %d" % (t[17] - t[9], t[10] - t[17]))
print("median deltas (cycles):", np.median(d.reshape(-1, 12), axis=0).astype(int))

order = [0, 1, 2, 3, 13, 14, 15, 16, 4, 5, 6, 7, 8, 9, 17, 10]
for itn in (0, 1):
    tt = np.median(ph[:, :, 32 + 32 * itn: 64 + 32 * itn].reshape(-1, 32), axis=0)
    t0 = np.median(ph[:, :, 32].reshape(-1))
    print("iteration", itn, "start at", int(tt[0] - t0), " ".join("%d:%d" % (k, tt[k] - tt[0]) for k in order))

# layer 2 in detail (last iteration), per wave of workgroup 0: 20 start, 21 after fc^T gemm, 22 after puts, 23 after barrier, 24 after FC tiles,
# 25 after barrier, 26 after puts (W phase), 27 after barrier, 28 after W tiles, 29 after barrier, 30 after W^T gemm
for w in range(8):
    t = ph[0, w]
    print("wg 0 wave", w, "layer 2:", " ".join("%d:%d" % (k, t[k] - t[20]) for k in range(20, 31)))
tm = np.median(ph[:, :, 20:31].reshape(-1, 11), axis=0)
print("median over 8 wgs x 8 waves:", " ".join("%d:%d" % (20 + k, tm[k] - tm[0]) for k in range(11)))
