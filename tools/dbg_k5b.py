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
o32, o64 = Oracle("f32"), Oracle("f64")
ro32, rd32 = o32.rays_from_pixels(pi, pj, *intr, o32.camera_from_tensor(cam0))
gmax = float(gt_d.max())
ctx = make_ctx(sc)
for k in (105, 184, 7):
    ro, rd, gd = ro32[k:k+1], rd32[k:k+1], gt_d[k:k+1]
    for tag, g_rgb, g_d in (("depth only", np.zeros((1, 3), np.float32), np.ones(1, np.float32)), ("colour only", np.ones((1, 3), np.float32), np.zeros(1, np.float32))):
        for stage in ("middle", "fine", "color"):
            if stage != "color" and tag == "colour only": continue
            g_ro, g_rd = ctx.render_backward(stage, cu(ro), cu(rd), cu(gd), gmax, cu(g_rgb), cu(g_d), None, flags=4)
            out = []
            for o in (o32, o64):
                bw = o.render_backward(o.opts(b), sc["grids"], sc["decoders"], stage, ro, rd, gd, gmax, g_rgb, g_d, None, want_grids=False, want_decoders=False)
                out.append(bw["g_rays_o"][0])
            print("ray", k, tag, stage, "hip", g_ro.cpu().numpy()[0], "f32", out[0], "f64", out[1])
print("---- forward of ray 105's samples")
for k in (105, 7):
    fw = o64.render_forward(o64.opts(b), sc["grids"], sc["decoders"], "color", ro32[k:k+1], rd32[k:k+1], gt_d[k:k+1], gmax, want_aux=True)
    fw32 = o32.render_forward(o32.opts(b), sc["grids"], sc["decoders"], "color", ro32[k:k+1], rd32[k:k+1], gt_d[k:k+1], gmax, want_aux=True)
    p = (ro32[k][None].astype(np.float32) + rd32[k][None].astype(np.float32) * fw32["z"][0][:, None].astype(np.float32)).astype(np.float32)
    raw = ctx.eval_points("color", cu(p)).cpu().numpy()
    d = np.abs(raw - fw["raw"][0]); d32 = np.abs(fw32["raw"][0] - fw["raw"][0])
    print("ray", k, "max |raw hip - f64| per channel", d.max(0), " f32 oracle:", d32.max(0), "at samples", d.argmax(0))
print("---- matmul mode 0 (fp32 MFMA forward)")
ctx0 = make_ctx(sc); ctx0.set_matmul_mode(0)
k = 105
g_ro, g_rd = ctx0.render_backward("fine", cu(ro32[k:k+1]), cu(rd32[k:k+1]), cu(gt_d[k:k+1]), gmax, cu(np.zeros((1,3),np.float32)), cu(np.ones(1,np.float32)), None, flags=4)
print("ray 105 depth only fine, mode 0: hip", g_ro.cpu().numpy()[0])
# which samples matter: zero out the ray beyond sample s by shortening? use eval of per-sample fragility through single-sample rays is not possible; print fragility per stage
for st in ("middle", "fine", "color"):
    print(st, "fragility of ray 105:", float(o64.ray_fragility(o64.opts(b), sc["grids"], sc["decoders"], st, ro32[k:k+1].astype(np.float64), rd32[k:k+1].astype(np.float64), gt_d[k:k+1], gmax)[0]),
          " in fp32:", float(o32.ray_fragility(o32.opts(b), sc["grids"], sc["decoders"], st, ro32[k:k+1], rd32[k:k+1], gt_d[k:k+1], gmax)[0]))
