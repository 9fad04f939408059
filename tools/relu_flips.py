"""experiment (round 4): where the all-rays gradient error comes from.  One colour-stage mapping step at a BASELINE config's full size
(default K3: 5000 rays x 48), for every forward matmul mode (0 fp32 MFMA, 1 three bf16 pieces, 2 two fp16 pieces):

  * the ReLU bits the forward saved (nsk_debug_relu_bits) and the ReLU inputs of the same bodies (nsk_debug_preact), per decoder, against the
    fp32 and the fp64 oracle (oracle/nso.c nso_preacts): branches that differ (counted per decoder and layer) and the rms / max input error;
  * the gradient of every trained level and of the colour decoder against (a) the fp32 oracle, (b) the fp64 oracle, (c) the fp32 / fp64
    oracle taking the branches the HIP forward took (nso_render_backward_forced): what is left when the kinks are out of the comparison;
  * which rays carry the difference between (b) and (c): the fp64 oracle's own-branch against forced-branch gradient, ray by ray.

usage: python tools/relu_flips.py [K2-color|K3-color|K3-fine|K4-shard] [modes, e.g. 012]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
from scenes import rel_l2
from gpu_util import cu, make_ctx
from oracle.nso import Oracle
import test_gpu_configs as tc

case = sys.argv[1] if len(sys.argv) > 1 else "K3-color"
modes = [int(ch) for ch in (sys.argv[2] if len(sys.argv) > 2 else "012")]
sc, rays, stage, gmax = tc._strict_case(case)
if gmax is None:
    gmax = float(rays["gt_depth"].max())
N = rays["rays_o"].shape[0]
S = 48
M = N * S
decs = list(tc.LEVELS[stage])                       # decoders of the stage = levels that receive gradient
names = decs + (["colour decoder"] if stage == "color" else [])
o32, o64 = Oracle("f32"), Oracle("f64")
W_COLOR = 0.5


def oracle_side(o, relu=None):
    op = o.opts(sc["bound"])
    fw = o.render_forward(op, sc["grids"], sc["decoders"], stage, rays["rays_o"], rays["rays_d"], rays["gt_depth"], gmax)
    l, g_d, g_c = o.loss_map(fw["depth"], fw["rgb"], rays["gt_depth"], rays["gt_color"], W_COLOR, stage == "color")
    bw = o.render_backward(op, sc["grids"], sc["decoders"], stage, rays["rays_o"], rays["rays_d"], rays["gt_depth"], gmax, g_c, g_d, None,
                           want_rays=False, relu=relu)
    return fw, bw, (g_d, g_c)


def pick(bw, k):
    return bw["g_decoders"]["color"] if k == "colour decoder" else bw["g_grids"][k]


print("== %s: %d rays x %d samples, stage %s ==" % (case, N, S, stage), flush=True)
fw32, bw32, _ = oracle_side(o32)
fw64, bw64, seeds64 = oracle_side(o64)
a32 = {k: o32.preacts(o32.opts(sc["bound"]), sc["grids"], sc["decoders"], stage, k, rays["rays_o"], rays["rays_d"], rays["gt_depth"], gmax) for k in decs}
a64 = {k: o64.preacts(o64.opts(sc["bound"]), sc["grids"], sc["decoders"], stage, k, rays["rays_o"], rays["rays_d"], rays["gt_depth"], gmax) for k in decs}
eo = {k: rel_l2(pick(bw32, k), pick(bw64, k)) for k in names}
print("fp32 oracle vs fp64 oracle, gradients: " + ", ".join("%s %.2e" % (k, eo[k]) for k in names))
for k in decs:
    d = a32[k].astype(np.float64) - a64[k]
    fl = (a32[k] > 0) != (a64[k] > 0)
    print("fp32 oracle vs fp64, decoder %-6s: ReLU inputs rms %.2e max %.2e | branches that differ per layer %s (of %d per layer)" % (
        k, np.sqrt((d ** 2).mean()), np.abs(d).max(), fl.sum(axis=(0, 2)).tolist(), M * 32))

ro, rd, gd, gc = cu(rays["rays_o"]), cu(rays["rays_d"]), cu(rays["gt_depth"]), cu(rays["gt_color"])
for mode in modes:
    ctx = make_ctx(sc, trainable=["color"] if stage == "color" else [])
    ctx.set_matmul_mode(mode)
    loss_t = torch.zeros(1, device="cuda")
    ctx.map_step(stage, ro, rd, gd, gc, gmax, W_COLOR, stage == "color", flags=3 if stage == "color" else 1, loss=loss_t)
    got = {k: (ctx.decoder_download("color", grad=True) if k == "colour decoder" else ctx.grid_download(k, grad=True)) for k in names}
    bits = {k: ctx.debug_relu_bits(k, M) for k in decs}
    pre = {k: ctx.debug_preact(k, ro, rd, M) for k in decs}
    print("-- forward matmul mode %d --" % mode)
    for k in decs:
        same_body = int(((pre[k] > 0) != bits[k]).sum())
        d64 = pre[k].astype(np.float64) - a64[k]
        d32 = pre[k].astype(np.float64) - a32[k]
        f64 = bits[k] != (a64[k] > 0)
        f32 = bits[k] != (a32[k] > 0)
        print("decoder %-6s: ReLU inputs vs fp64 rms %.2e max %.2e, vs fp32 oracle rms %.2e max %.2e | branches vs fp64 %s = %d, vs fp32 oracle %s = %d | "
              "largest |input| (fp64) at a differing branch %.2e | dump kernel vs saved bits: %d differ" % (
                  k, np.sqrt((d64 ** 2).mean()), np.abs(d64).max(), np.sqrt((d32 ** 2).mean()), np.abs(d32).max(),
                  f64.sum(axis=(0, 2)).tolist(), f64.sum(), f32.sum(axis=(0, 2)).tolist(), f32.sum(),
                  np.abs(a64[k][f64]).max() if f64.any() else 0.0, same_body))
    _, bw32f, _ = oracle_side(o32, relu=bits)
    _, bw64f, _ = oracle_side(o64, relu=bits)
    for k in names:
        print("d loss / d %-14s: hip vs fp32 oracle %.2e, vs fp64 %.2e (oracles %.2e apart) | branches forced to the HIP forward's: vs fp32 oracle %.2e, vs fp64 %.2e" % (
            k, rel_l2(got[k], pick(bw32, k)), rel_l2(got[k], pick(bw64, k)), eo[k], rel_l2(got[k], pick(bw32f, k)), rel_l2(got[k], pick(bw64f, k))))
    # which rays carry the own-branch / forced-branch difference of the fp64 oracle (the kink part of the error)?
    flipped = np.zeros(N, bool)
    for k in decs:
        flipped |= (bits[k] != (a64[k] > 0)).reshape(N, S, -1).any(axis=(1, 2))
    idx = np.nonzero(flipped)[0]
    print("rays with a branch that differs from the fp64 oracle: %d of %d" % (idx.size, N))
    if idx.size and stage == "color":
        g_d, g_c = seeds64
        op = o64.opts(sc["bound"])
        contrib = []
        for n in idx:
            sl = slice(n, n + 1)
            rb = {k: bits[k].reshape(N, S, 5, 32)[n].reshape(S, 5, 32) for k in decs}
            own = o64.render_backward(op, sc["grids"], sc["decoders"], stage, rays["rays_o"][sl], rays["rays_d"][sl], rays["gt_depth"][sl], gmax,
                                      g_c[sl], g_d[sl], None, want_rays=False, want_decoders=False)
            frc = o64.render_backward(op, sc["grids"], sc["decoders"], stage, rays["rays_o"][sl], rays["rays_d"][sl], rays["gt_depth"][sl], gmax,
                                      g_c[sl], g_d[sl], None, want_rays=False, want_decoders=False, relu=rb)
            contrib.append([float(((own["g_grids"][k] - frc["g_grids"][k]) ** 2).sum()) for k in decs])
        contrib = np.array(contrib)
        for j, k in enumerate(decs):
            c = np.sort(contrib[:, j])[::-1]
            tot = c.sum()
            nrm2 = float((pick(bw64, k) ** 2).sum())
            print("level %-6s: kink part of the squared error %.3e = (%.2e)^2 of the gradient; top rays' shares %s; rays for 90 %%: %d" % (
                k, tot, np.sqrt(tot / nrm2), ["%.2f" % (x / tot) for x in c[:5]], int(np.searchsorted(np.cumsum(c), 0.9 * tot) + 1)))
    ctx.close()
