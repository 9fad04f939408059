import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
from oracle.nso import Oracle
from gpu_util import cu, make_ctx
from test_gpu_configs import _quat_cam
cam = scenes.CAM_TUM; intr = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
sc = scenes.make_scene(81, scenes.grid_shapes_for(scenes.K5_BOUND), bound=scenes.K5_BOUND, grid_std=0.2, bias_std=0.05)
b = sc["bound"]
r = scenes.make_rays(83, 500, b, n_frames=1, edge=20, up="z", **cam)
tsel = np.arange(0, 500, 2)[:200]
cam0 = _quat_cam(r["c2w"][0], 1.2, (0.015, -0.01, 0.02))
pi, pj, gt_d, gt_c = r["pix_i"][tsel], r["pix_j"][tsel], r["gt_depth"][tsel], r["gt_color"][tsel]
ctx = make_ctx(sc)
cam_t = cu(cam0)
ro, rd = ctx.rays_from_camera(cu(pi, torch.int32), cu(pj, torch.int32), intr, cam_t)
g_ro = torch.empty_like(ro); g_rd = torch.empty_like(rd); loss = torch.zeros(1, device="cuda")
ctx.track_step("color", ro, rd, cu(gt_d), cu(gt_c), -1.0, 0.5, True, True, True, flags=4, loss=loss, g_rays=(g_ro, g_rd))
g_c2w = ctx.rays_backward(cu(pi, torch.int32), cu(pj, torch.int32), intr, g_ro, g_rd)
g_hip = ctx.camera_backward(cam_t, g_c2w).cpu().numpy()
res = {}
for name in ("f32", "f64"):
    o = Oracle(name); o.lib.nso_set_num_threads(16)
    c = cam0.astype(o.dt)
    c2w = o.camera_from_tensor(c); ro_, rd_ = o.rays_from_pixels(pi, pj, *intr, c2w)
    op = o.opts(b)
    fw = o.render_forward(op, sc["grids"], sc["decoders"], "color", ro_, rd_, gt_d)
    l, gD, gC, gV = o.loss_track(fw["depth"], fw["rgb"], fw["var"], gt_d, gt_c, 0.5, True, True, True)
    bw = o.render_backward(op, sc["grids"], sc["decoders"], "color", ro_, rd_, gt_d, -1.0, gC, gD, None, want_grids=False, want_decoders=False)
    res[name] = (o.camera_backward(c, o.rays_backward(pi, pj, *intr, bw["g_rays_o"], bw["g_rays_d"])), l, bw, gD, fw)
print("loss hip %.6f f32 %.6f f64 %.6f" % (float(loss), res["f32"][1], res["f64"][1]))
print("g_cam hip", g_hip); print("g_cam f32", res["f32"][0]); print("g_cam f64", res["f64"][0])
print("rel: hip-f32 %.2e hip-f64 %.2e f32-f64 %.2e" % (scenes.rel_l2(g_hip, res["f32"][0]), scenes.rel_l2(g_hip, res["f64"][0]), scenes.rel_l2(res["f32"][0], res["f64"][0])))
gro, grd = g_ro.cpu().numpy(), g_rd.cpu().numpy()
e = np.abs(gro - res["f32"][2]["g_rays_o"]).sum(1); i = np.argsort(e)[::-1][:5]
print("rays_o rel %.2e rays_d rel %.2e" % (scenes.rel_l2(gro, res["f32"][2]["g_rays_o"]), scenes.rel_l2(grd, res["f32"][2]["g_rays_d"])))
for k in i: print(k, gro[k], res["f32"][2]["g_rays_o"][k], res["f64"][2]["g_rays_o"][k], "gD", res["f32"][3][k], res["f64"][3][k], "depth", res["f32"][4]["depth"][k], gt_d[k])
o64 = Oracle("f64")
c = cam0.astype(np.float64); c2w = o64.camera_from_tensor(c); ro_, rd_ = o64.rays_from_pixels(pi, pj, *intr, c2w)
frag = o64.ray_fragility(o64.opts(b), sc["grids"], sc["decoders"], "color", ro_, rd_, gt_d)
print("fragility of the worst rays:", [(int(k), float(frag[k])) for k in i], "median", float(np.median(frag)), "share < 2e-5: %.2f" % (frag < 2e-5).mean())
keep = frag > 2e-5
print("rays_o rel on non-fragile rays %.2e (n=%d), on fragile %.2e" % (scenes.rel_l2(gro[keep], res["f32"][2]["g_rays_o"][keep]), keep.sum(), scenes.rel_l2(gro[~keep], res["f32"][2]["g_rays_o"][~keep])))
# same step with the ORACLE's fp32 rays fed to the GPU (no device-side ray generation)
o32 = Oracle("f32"); c32 = cam0.astype(np.float32)
ro32, rd32 = o32.rays_from_pixels(pi, pj, *intr, o32.camera_from_tensor(c32))
print("device rays vs oracle-f32 rays: rays_d max abs diff %.3e (ulp-level?), rays_o %.3e" % (np.abs(rd.cpu().numpy() - rd32).max(), np.abs(ro.cpu().numpy() - ro32).max()))
g_ro2 = torch.empty_like(ro); g_rd2 = torch.empty_like(rd)
ctx.track_step("color", cu(ro32), cu(rd32), cu(gt_d), cu(gt_c), -1.0, 0.5, True, True, True, flags=4, loss=loss, g_rays=(g_ro2, g_rd2))
print("with oracle rays: rays_o rel %.2e rays_d rel %.2e" % (scenes.rel_l2(g_ro2.cpu().numpy(), res["f32"][2]["g_rays_o"]), scenes.rel_l2(g_rd2.cpu().numpy(), res["f32"][2]["g_rays_d"])))
print("---- pose gradient by forward matmul mode")
for mode in (0, 1):
    c2 = make_ctx(sc); c2.set_matmul_mode(mode)
    gro_ = torch.empty_like(ro); grd_ = torch.empty_like(rd)
    c2.track_step("color", cu(ro32), cu(rd32), cu(gt_d), cu(gt_c), -1.0, 0.5, True, True, True, flags=4, loss=loss, g_rays=(gro_, grd_))
    g = c2.camera_backward(cam_t, c2.rays_backward(cu(pi, torch.int32), cu(pj, torch.int32), intr, gro_, grd_)).cpu().numpy()
    bad = (np.abs(gro_.cpu().numpy() - res["f64"][2]["g_rays_o"]).max(1) > 1e-2 * np.abs(res["f64"][2]["g_rays_o"]).max(1)).sum()
    print("mode", mode, "g_cam rel: vs f32 %.2e vs f64 %.2e; rays off by > 1 %%: %d of %d" % (scenes.rel_l2(g, res["f32"][0]), scenes.rel_l2(g, res["f64"][0]), bad, len(pi)))
bad32 = (np.abs(res["f32"][2]["g_rays_o"] - res["f64"][2]["g_rays_o"]).max(1) > 1e-2 * np.abs(res["f64"][2]["g_rays_o"]).max(1)).sum()
print("f32 oracle vs f64: rays off by > 1 %%: %d" % bad32)
