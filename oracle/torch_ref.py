"""Second, independent CPU restatement of the hot path, built from the SAME ATen ops the reference
calls (F.grid_sample, F.linear, matmul+sin, sort, cumprod, autograd, torch.optim.Adam), op for op in
the order of the reference sources.  TEST INFRASTRUCTURE ONLY (see oracle/nso.c header).

It exists to pin oracle/nso.c (analytic backward, plain C) against autograd over the reference's own
op sequence, and to generate tests/golden/*.npz (tests/golden/make_golden.py).  Intended-vs-as-written
decisions follow SURVEY.md section 0.3.
"""
import torch
import torch.nn.functional as F

E_DIM, H_DIM = 93, 32
LEVELS = ("coarse", "middle", "fine", "color")


def decoder_layout(which):
    """offsets into the packed parameter vector (same packing as oracle/nso.c make_layout)"""
    has_xyz = which != "coarse"
    c_dim = 64 if which == "fine" else 32
    out_dim = 4 if which == "color" else 1
    in_dims = [E_DIM, H_DIM, H_DIM, H_DIM + E_DIM, H_DIM] if has_xyz else [32, H_DIM, H_DIM, H_DIM + 32, H_DIM]
    o = 0
    lay = dict(has_xyz=has_xyz, c_dim=c_dim, out_dim=out_dim, in_dims=in_dims)
    if has_xyz:
        lay["B"] = (o, (3, E_DIM)); o += 3 * E_DIM
    lay["W"], lay["b"], lay["Fw"], lay["Fb"] = [], [], [], []
    for i in range(5):
        lay["W"].append((o, (H_DIM, in_dims[i]))); o += H_DIM * in_dims[i]
        lay["b"].append((o, (H_DIM,))); o += H_DIM
    if has_xyz:
        for i in range(5):
            lay["Fw"].append((o, (H_DIM, c_dim))); o += H_DIM * c_dim
            lay["Fb"].append((o, (H_DIM,))); o += H_DIM
    lay["Wo"] = (o, (out_dim, H_DIM)); o += out_dim * H_DIM
    lay["bo"] = (o, (out_dim,)); o += out_dim
    lay["total"] = o
    return lay


def _view(P, ent):
    o, shp = ent
    n = 1
    for s in shp:
        n *= s
    return P[o:o + n].view(*shp)


def init_decoder(which, gen):
    """src/models/MLP.cpp:65-74 (xavier_uniform with ReLU gain, zero bias), GaussianFFT.cpp:6 (randn*25).
    fc layers keep torch::nn::Linear's default init (kaiming_uniform(a=sqrt(5)) -> U(-1/sqrt(in),1/sqrt(in)))."""
    lay = decoder_layout(which)
    P = torch.zeros(lay["total"])
    gain = 2.0 ** 0.5
    if lay["has_xyz"]:
        _view(P, lay["B"]).copy_(torch.randn(3, E_DIM, generator=gen) * 25)
    for i in range(5):
        fo, fi = H_DIM, lay["in_dims"][i]
        a = gain * (6.0 / (fi + fo)) ** 0.5
        _view(P, lay["W"][i]).copy_((torch.rand(fo, fi, generator=gen) * 2 - 1) * a)
    if lay["has_xyz"]:
        for i in range(5):
            k = 1.0 / lay["c_dim"] ** 0.5
            _view(P, lay["Fw"][i]).copy_((torch.rand(H_DIM, lay["c_dim"], generator=gen) * 2 - 1) * k)
            _view(P, lay["Fb"][i]).copy_((torch.rand(H_DIM, generator=gen) * 2 - 1) * k)
    fo, fi = lay["out_dim"], H_DIM
    a = gain * (6.0 / (fi + fo)) ** 0.5
    _view(P, lay["Wo"]).copy_((torch.rand(fo, fi, generator=gen) * 2 - 1) * a)
    return P


def sample_grid_feature(p, grid, bound):
    """MLP::sample_grid_feature src/models/MLP.cpp:51-63 with normalize_3d_coordinate utils.h:132-139
    (intended: normalise a copy, return the sampled features; D12, D13)."""
    p_nor = torch.stack([((p[:, k] - bound[k, 0]) / (bound[k, 1] - bound[k, 0])) * 2 - 1 for k in range(3)], -1)
    vgrid = p_nor.unsqueeze(0)[:, :, None, None]                                   # [1,M,1,1,3]
    c = F.grid_sample(grid, vgrid, mode="bilinear", padding_mode="border", align_corners=True)
    return c.squeeze(-1).squeeze(-1)                                                # [1,C,M]


def decoder_forward(which, P, p, grids, bound):
    """MLP::forward src/models/MLP.cpp:76-102 / MLP_no_xyz::forward :165-182 (intended loops D14, D15)."""
    lay = decoder_layout(which)
    c = sample_grid_feature(p, grids[which], bound).transpose(1, 2).squeeze(0)      # [M,C]
    if which == "fine":
        with torch.no_grad():                                                        # MLP.cpp:81
            c_middle = sample_grid_feature(p, grids["middle"], bound).transpose(1, 2).squeeze(0)
        c = torch.cat([c, c_middle], 1)
    if lay["has_xyz"]:
        embedded = torch.sin(torch.matmul(p, _view(P, lay["B"])))                   # GaussianFFT.cpp:13-14
        h = embedded
    else:
        embedded = None
        h = c
    for i in range(5):
        h = F.relu(F.linear(h, _view(P, lay["W"][i]), _view(P, lay["b"][i])))
        if lay["has_xyz"]:
            h = h + F.linear(c, _view(P, lay["Fw"][i]), _view(P, lay["Fb"][i]))
        if i == 2:
            h = torch.cat([embedded if lay["has_xyz"] else c, h], -1)
    out = F.linear(h, _view(P, lay["Wo"]), _view(P, lay["bo"]))
    if lay["out_dim"] == 1:
        out = out.squeeze(-1)
    return out


def nice_forward(stage, decoders, p, grids, bound):
    """NICE::forward src/models/NICE.cpp:16-52"""
    M = p.shape[0]
    if stage in ("coarse", "middle"):
        occ = decoder_forward(stage, decoders[stage], p, grids, bound)
        return torch.cat([torch.zeros(M, 3), occ[:, None]], -1)
    fine_occ = decoder_forward("fine", decoders["fine"], p, grids, bound)
    middle_occ = decoder_forward("middle", decoders["middle"], p, grids, bound)
    if stage == "fine":
        return torch.cat([torch.zeros(M, 3), (fine_occ + middle_occ)[:, None]], -1)
    raw = decoder_forward("color", decoders["color"], p, grids, bound)
    return torch.cat([raw[:, :3], (fine_occ + middle_occ)[:, None]], -1)


def eval_points(p, stage, decoders, grids, bound, points_batch_size=100000):
    """Renderer::eval_points src/Renderer.cpp:19-42"""
    rets = []
    for pi in torch.split(p, points_batch_size):
        mask_x = (pi[:, 0] < bound[0, 1]) & (pi[:, 0] > bound[0, 0])
        mask_y = (pi[:, 1] < bound[1, 1]) & (pi[:, 1] > bound[1, 0])
        mask_z = (pi[:, 2] < bound[2, 1]) & (pi[:, 2] > bound[2, 0])
        mask = mask_x & mask_y & mask_z
        ret = nice_forward(stage, decoders, pi, grids, bound)
        occ = torch.where(mask, ret[:, 3], torch.full_like(ret[:, 3], 100.0))       # ret[~mask,3]=100
        rets.append(torch.cat([ret[:, :3], occ[:, None]], -1))
    return torch.cat(rets, 0)


def raw2outputs_nerf_color(raw, z_vals, occupancy, rays_d):
    """include/torchlib/utils.h:148-172"""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.tensor([1e10]).expand(dists[..., :1].shape)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = raw[..., :-1]
    if occupancy:
        alpha = torch.sigmoid(10 * raw[..., -1])
    else:
        alpha = 1 - torch.exp(-F.relu(raw[..., -1]) * dists)
    weights = alpha * torch.cumprod(torch.cat([torch.ones(alpha.shape[0], 1), 1 - alpha + 1e-10], -1), -1)[:, :-1]
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    tmp = z_vals - depth_map.unsqueeze(-1)
    depth_var = torch.sum(weights * tmp * tmp, 1)
    return rgb_map, depth_map, depth_var, weights


def render_batch_ray(grids, decoders, rays_d, rays_o, stage, gt_depth, bound, n_samples=32, n_surface=16,
                     lindisp=False, occupancy=False, gt_depth_max=None, return_aux=False):
    """Renderer::render_batch_ray src/Renderer.cpp:44-126 (perturb=0).  gt_depth may be None."""
    N = rays_o.shape[0]
    if gt_depth is None:
        n_surface = 0                                                                # :56 (D8: local)
        near = torch.tensor([0.01])
    else:
        gt_depth = gt_depth.reshape(-1, 1)
        near = gt_depth.repeat(1, n_samples) * 0.01                                  # :62-63
    with torch.no_grad():                                                            # :66-73 (D6)
        det_rays_o = rays_o.detach().unsqueeze(-1)
        det_rays_d = rays_d.detach().unsqueeze(-1)
        t = (bound.unsqueeze(0) - det_rays_o) / det_rays_d
        far_bb, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
        far_bb = far_bb.unsqueeze(-1) + 0.01
    if gt_depth is not None:
        gmax = torch.max(gt_depth) if gt_depth_max is None else torch.tensor(float(gt_depth_max))
        far = torch.clamp(far_bb, torch.tensor(0.0), gmax * 1.2)                     # :76
    else:
        far = far_bb
    if n_surface > 0:                                                                # :80-99
        nz = (gt_depth > 0).squeeze(-1)
        t_surf = torch.linspace(0, 1, n_surface)
        z_surface = torch.zeros(N, n_surface)
        g = gt_depth[nz].reshape(-1, 1).repeat(1, n_surface)
        z_surface[nz] = 0.95 * g * (1 - t_surf) + 1.05 * g * t_surf
        far_surface = torch.max(gt_depth) if gt_depth_max is None else torch.tensor(float(gt_depth_max))
        z_zero = torch.tensor([0.001]) * (1. - t_surf) + far_surface * t_surf
        z_surface[~nz] = z_zero.unsqueeze(0).repeat(int((~nz).sum()), 1)
    t_vals = torch.linspace(0, 1, n_samples)                                         # :101
    if not lindisp:
        z_vals = near * (1 - t_vals) + far * t_vals
    else:
        z_vals = 1. / (1. / near * (1. - t_vals) + 1. / far * t_vals)
    if n_surface > 0:
        z_vals, _ = torch.sort(torch.cat([z_vals, z_surface], -1), -1)               # :119
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]         # :121
    raw = eval_points(pts.reshape(-1, 3), stage, decoders, grids, bound)
    raw = raw.reshape(N, n_samples + n_surface, -1)
    out = raw2outputs_nerf_color(raw, z_vals, occupancy, rays_d)
    if return_aux:
        return out + (z_vals, raw)
    return out


def loss_map(depth, color, gt_depth, gt_color, w_color, use_color):
    """src/Mapper.cpp:435-442"""
    depth_mask = gt_depth > 0
    loss = torch.abs(gt_depth[depth_mask] - depth[depth_mask]).sum()
    if use_color:
        loss = loss + w_color * torch.abs(gt_color - color).sum()
    return loss


def loss_track(depth, color, var, gt_depth, gt_color, w_color, use_color, handle_dynamic, detach_var=True):
    """src/Tracker.cpp:67-82"""
    if detach_var:
        var = var.detach()
    if handle_dynamic:
        tmp = torch.abs(gt_depth - depth).detach()
        mask = (tmp < 10 * tmp.median()) & (gt_depth > 0)
    else:
        mask = gt_depth > 0
    loss = (torch.abs(gt_depth - depth) / torch.sqrt(var + 1e-10))[mask].sum()
    if use_color:
        loss = loss + w_color * torch.abs(gt_color - color)[mask].sum()
    return loss


def quad2rotation(quad):
    """include/torchlib/utils.h:174-195 (written out-of-place so autograd can differentiate it)"""
    qr, qi, qj, qk = quad[:, 0], quad[:, 1], quad[:, 2], quad[:, 3]
    two_s = 2 / (quad * quad).sum(-1)
    rows = [
        torch.stack([1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr)], -1),
        torch.stack([two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr)], -1),
        torch.stack([two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2)], -1)]
    return torch.stack(rows, 1)


def get_camera_from_tensor(inputs):
    """include/torchlib/utils.h:198-210 for a single 7-vector (qw,qx,qy,qz,tx,ty,tz) -> [3,4]"""
    quad, T = inputs[None, :4], inputs[None, 4:]
    R = quad2rotation(quad)
    return torch.cat([R, T[:, :, None]], 2)[0]


def rays_from_pixels(pix_i, pix_j, fx, fy, cx, cy, c2w, mode=0):
    """raySampler include/torchlib/utils.h:44-52 with the pixel choice as an input (intended dirs, D11)."""
    if mode & 2:
        fx, fy, cx, cy = float(int(fx)), float(int(fy)), float(int(cx)), float(int(cy))
    i, j = pix_i.to(c2w.dtype), pix_j.to(c2w.dtype)
    i_t = (i - cx) / fx
    j_t = (i - cy) / fy if (mode & 1) else -(j - cy) / fy
    dirs = torch.stack([i_t, j_t, -torch.ones_like(i)], -1).reshape(-1, 1, 3)
    rays_d = torch.sum(dirs * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d
