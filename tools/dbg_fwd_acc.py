import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
from oracle.nso import Oracle
from gpu_util import cu, make_ctx
sc = scenes.make_scene(81, scenes.grid_shapes_for(scenes.K5_BOUND), bound=scenes.K5_BOUND, grid_std=0.2, bias_std=0.05)
b = sc["bound"]
r = scenes.make_rays(83, 400, b, n_frames=1, edge=20, up="z", **scenes.CAM_TUM)
o32, o64 = Oracle("f32"), Oracle("f64")
fw64 = o64.render_forward(o64.opts(b), sc["grids"], sc["decoders"], "color", r["rays_o"], r["rays_d"], r["gt_depth"], want_aux=True)
fw32 = o32.render_forward(o32.opts(b), sc["grids"], sc["decoders"], "color", r["rays_o"], r["rays_d"], r["gt_depth"], want_aux=True)
# identical fp32 points for everybody: the fp32 oracle's own p = o + d z
p = (r["rays_o"][:, None, :] + r["rays_d"][:, None, :] * fw32["z"][:, :, None]).astype(np.float32).reshape(-1, 3)
ref32 = fw32["raw"].reshape(-1, 4)
inb = np.all((p < b[:, 1]) & (p > b[:, 0]), axis=1)
for mode in (0, 1):
    ctx = make_ctx(sc); ctx.set_matmul_mode(mode)
    raw = ctx.eval_points("color", cu(p)).cpu().numpy()
    d = np.abs(raw - ref32)[inb]
    print("mode %d vs fp32 oracle (same fp32 points): max %.2e  99.9%% %.2e  median %.2e  rms %.2e" % (mode, d.max(), np.quantile(d, 0.999), np.median(d), np.sqrt((d**2).mean())))
