"""per-kernel HIP-event times of one Tracker iteration (Tracker::optimize_cam_in_batch, 200 rays): python tools/track_times.py [rays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nice_slam_cpp_amd as pkg, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sc = scenes.make_scene(42)
r = scenes.make_rays(7, N, sc["bound"], H=680, W=1200, n_frames=1, edge=20)
cu = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"])
for kv in sys.argv[2:]:          # key=value pairs for nsk_set_tuning (e.g. no_fused_median=1)
    k, v_ = kv.split("="); ctx.set_tuning(k, int(v_))
c2w = r["c2w"][0]
R = c2w[:3, :3].astype(np.float64); qw = np.sqrt(max(1e-12, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
q = np.array([qw, (R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw)])
cam = cu(np.concatenate([q, c2w[:3, 3] + 0.01]).astype(np.float32)); m = torch.zeros(7, device="cuda"); v = torch.zeros(7, device="cuda")
pi, pj = cu(r["pix_i"], torch.int32), cu(r["pix_j"], torch.int32)
gd, gc = cu(r["gt_depth"]), cu(r["gt_color"])
loss = torch.zeros(1, device="cuda")
g_ro = torch.empty(N, 3, device="cuda"); g_rd = torch.empty(N, 3, device="cuda")
def it(step):
    c = ctx.camera_from_tensor(cam)
    ro, rd = ctx.rays_from_pixels(pi, pj, r["intr"], c)
    ctx.track_step("color", ro, rd, gd, gc, -1.0, 0.5, True, True, True, flags=4, loss=loss, g_rays=(g_ro, g_rd))
    g_c2w = ctx.rays_backward(pi, pj, r["intr"], g_ro, g_rd)
    g_cam = ctx.camera_backward(cam, g_c2w)
    ctx.adam_vector(cam, g_cam, m, v, 1e-3, step)
with torch.cuda.stream(ctx.tstream):
    for s in range(1, 6): it(s)
    ctx.profile_begin()
    for s in range(6, 26): it(s)
    p = ctx.profile_end()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for s in range(26, 126): it(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    # fused pose kernels: 8 launches instead of 11
    def it_fused(step):
        ro, rd = ctx.rays_from_camera(pi, pj, r["intr"], cam)
        ctx.track_step("color", ro, rd, gd, gc, -1.0, 0.5, True, True, True, flags=4, loss=loss, g_rays=(g_ro, g_rd))
        ctx.pose_step(pi, pj, r["intr"], g_ro, g_rd, cam, m, v, 1e-3, step)
    for s in range(200, 210): it_fused(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(210, 310): it_fused(s)
    torch.cuda.synchronize()
    dtf = (time.perf_counter() - t0) / 100
    # the same iteration as one captured hipGraph (tensors created while capturing are kept alive: their addresses are recorded)
    keep = []
    def it_keep(step):
        c = ctx.camera_from_tensor(cam)
        ro, rd = ctx.rays_from_pixels(pi, pj, r["intr"], c)
        ctx.track_step("color", ro, rd, gd, gc, -1.0, 0.5, True, True, True, flags=4, loss=loss, g_rays=(g_ro, g_rd))
        g_c2w = ctx.rays_backward(pi, pj, r["intr"], g_ro, g_rd)
        g_cam = ctx.camera_backward(cam, g_c2w)
        ctx.adam_vector(cam, g_cam, m, v, 1e-3, step)
        keep.extend([c, ro, rd, g_c2w, g_cam])
    ctx.graph_begin(); it_keep(126); gid = ctx.graph_end()
    for s in range(10): ctx.graph_launch(gid)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(100): ctx.graph_launch(gid)
    torch.cuda.synchronize()
    dtg = (time.perf_counter() - t0) / 100
d = {k: round(1e3 * ms / c, 1) for k, (c, ms) in p.items()}
print("graph replay: wall per iteration %.1f us; fused pose kernels: %.1f us; loss %.4f" % (dtg * 1e6, dtf * 1e6, float(loss)))
print("rays", N, "wall per iteration %.1f us" % (dt * 1e6), "profiled kernels sum %.1f us" % sum(d.values()), d, "loss %.4f" % float(loss))
