"""CPU test (-m "not gpu") of the N > 1 path: world size 2 over gloo.  The sharding / all-reduce host logic
(nice-slam-cpp_amd/dist.py, the code bench.py runs per rank) is driven with the CPU oracle standing in for the HIP
context: two ranks on ray shards + one all-reduce per step must reproduce the single-process full-batch step."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """same interface as nice_slam_cpp_amd.Context for the calls ShardedMapper makes (test stand-in)"""

    def __init__(self, sc):
        from oracle.nso import Oracle
        self.o = Oracle("f64")
        self.bound = sc["bound"]
        self.grids = {k: v.astype(np.float64) for k, v in sc["grids"].items()}
        self.decs = {k: v.astype(np.float64) for k, v in sc["decoders"].items()}
        self.levels = ("middle", "fine", "color")
        sizes = [self.grids[k].size for k in self.levels] + [self.decs["color"].size, 1]
        self.off = np.concatenate([[0], np.cumsum(sizes)])
        self.slab = torch.zeros(int(self.off[-1]), dtype=torch.float64)
        self.mom = {k: (np.zeros_like(self.grids[k]), np.zeros_like(self.grids[k])) for k in self.levels}
        self.dm, self.dv = np.zeros_like(self.decs["color"]), np.zeros_like(self.decs["color"])
        self.t = 0

    def grad_slab(self):
        return self.slab

    # the compact exchange of nsk_grad_pack / nsk_grad_unpack: marked voxels only (self.masks: level -> bool [Z,Y,X] or None)
    masks = None

    def _sel(self, i, k):
        if self.masks is None or self.masks.get(k) is None:
            return None
        return np.flatnonzero(np.broadcast_to(self.masks[k][None], self.grids[k].shape).ravel())

    def grad_pack(self):
        s = self.slab.numpy()
        parts = []
        for i, k in enumerate(self.levels):
            seg = s[self.off[i]:self.off[i + 1]]
            sel = self._sel(i, k)
            parts.append(seg if sel is None else seg[sel])
        parts.append(s[self.off[3]:])
        if self.extra is not None:
            parts.append(np.asarray(self.extra, np.float64))             # nsk_grad_extra: travels behind the loss floats
        self._packed = torch.tensor(np.concatenate(parts))
        return self._packed

    def grad_unpack(self):
        s = self.slab.numpy()
        p = self._packed.numpy()
        o = 0
        for i, k in enumerate(self.levels):
            sel = self._sel(i, k)
            n = (self.off[i + 1] - self.off[i]) if sel is None else sel.size
            if sel is None:
                s[self.off[i]:self.off[i + 1]] = p[o:o + n]
            else:
                s[self.off[i]:self.off[i + 1]][sel] = p[o:o + n]
            o += n
        tail = s.size - self.off[3]
        s[self.off[3]:] = p[o:o + tail]
        if self.extra is not None:
            self.extra[:] = p[o + tail:]

    def map_step(self, stage, ro, rd, gd, gc, gmax, w_color, use_color, flags=3, loss=None, g_rays=None):
        if g_rays is not None:
            return self._map_step_ba(stage, ro, rd, gd, gc, w_color, use_color, loss, g_rays)
        o, op = self.o, self.o.opts(self.bound)
        fw = o.render_forward(op, self.grids, self.decs, stage, ro, rd, gd, gt_depth_max=gmax)
        l, g_d, g_c = o.loss_map(fw["depth"], fw["rgb"], gd, gc, w_color, use_color)
        bw = o.render_backward(op, self.grids, self.decs, stage, ro, rd, gd, gmax, g_c, g_d, None, want_rays=False)
        s = self.slab.numpy()
        for i, k in enumerate(self.levels):
            s[self.off[i]:self.off[i + 1]] += bw["g_grids"][k].ravel()
        s[self.off[3]:self.off[4]] += bw["g_decoders"]["color"]
        s[self.off[4]] += l

    # ---- bundle adjustment (ShardedMapper.step_ba): the calls of nsk.h with the same meaning ------------------------------------------
    dmax = None; keep = None; extra = None

    def set_depth_max_batch(self, gt_depth, keep):
        self.dmax = None if gt_depth is None else (np.asarray(gt_depth), None if keep is None else np.asarray(keep))

    def set_ray_mask(self, keep):
        self.keep = None if keep is None else np.asarray(keep).astype(bool)

    def grad_extra(self, buf):
        self.extra = buf

    def _map_step_ba(self, stage, ro, rd, gd, gc, w_color, use_color, loss, g_rays):
        o, op = self.o, self.o.opts(self.bound)
        if self.dmax is not None:                                       # nsk_set_depth_max_batch: the maximum of the batch this shard belongs to
            g, k = self.dmax
            gmax = float(g[k.astype(bool)].max()) if k is not None else float(g.max())
        else:
            gmax = float(np.asarray(gd)[self.keep].max()) if self.keep is not None else float(np.max(gd))
        fw = o.render_forward(op, self.grids, self.decs, stage, ro, rd, gd, gt_depth_max=gmax)
        l, g_d, g_c = o.loss_map(fw["depth"], fw["rgb"], gd, gc, w_color, use_color)
        if self.keep is not None:                                       # nsk_set_ray_mask: masked rays carry neither loss nor gradient
            per = np.abs(np.asarray(gd) - fw["depth"]) * (np.asarray(gd) > 0) + (w_color * np.abs(np.asarray(gc) - fw["rgb"]).sum(1) if use_color else 0)
            l = float(per[self.keep].sum())
            g_d = g_d * self.keep; g_c = g_c * self.keep[:, None]
        bw = o.render_backward(op, self.grids, self.decs, stage, ro, rd, gd, gmax, g_c, g_d, None, want_rays=True)
        s = self.slab.numpy()
        for i, k in enumerate(self.levels):
            s[self.off[i]:self.off[i + 1]] += bw["g_grids"][k].ravel()
        s[self.off[3]:self.off[4]] += bw["g_decoders"]["color"]
        loss[0] = l
        g_rays[0][:] = bw["g_rays_o"]; g_rays[1][:] = bw["g_rays_d"]

    def pose_step_multi(self, first, count, active, pix_i, pix_j, intr, g_ro, g_rd, cams, step=0, mode=0, g_cams=None, keep=None):
        assert step == 0
        nf = len(first)
        g_cams[:] = 0
        for f in range(nf):
            if not active[f] or count[f] == 0:
                continue
            sl = slice(first[f], first[f] + count[f])
            g_c2w = self.o.rays_backward(pix_i[sl], pix_j[sl], *intr, g_ro[sl], g_rd[sl], mode=mode)
            g_cams[8 * f:8 * f + 7] = self.o.camera_backward(cams[f, :7], g_c2w)
        g_cams[8 * nf + 1] = float(np.asarray(keep).astype(bool).sum()) if keep is not None else 0.0

    def adam_vector(self, p, g, m, v, lr, step):
        self.o.adam_step(p, np.ascontiguousarray(g), m, v, lr, step)

    def adam_step(self, lr):
        self.t += 1
        s = self.slab.numpy()
        for i, k in enumerate(self.levels):
            g = s[self.off[i]:self.off[i + 1]].reshape(self.grids[k].shape)
            vm = None if (self.masks is None or self.masks.get(k) is None) else np.broadcast_to(self.masks[k][None], self.grids[k].shape)
            self.o.adam_step(self.grids[k], g, self.mom[k][0], self.mom[k][1], lr[2 + i], self.t, mask=vm)
        self.o.adam_step(self.decs["color"], s[self.off[3]:self.off[4]].copy(), self.dm, self.dv, lr[0], self.t)
        self.loss = float(s[self.off[4]])
        s[:] = 0


def _scene_and_rays(kind="small"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes
    if kind == "k4":                                                 # BASELINE configs[3] at its per-rank size: office0-class grids, two 1250-ray shards
        sc = scenes.make_scene(71, scenes.grid_shapes_for(scenes.K4_BOUND), bound=scenes.K4_BOUND)
        return sc, scenes.make_rays(72, 2500, sc["bound"], n_frames=5, up="z", **scenes.CAM_NICE_SLAM)
    sc = scenes.make_scene(5, scenes.SMALL_GRID_SHAPES, grid_std=0.1)
    rays = scenes.make_rays(6, 37, sc["bound"], n_frames=1)          # odd count: uneven shards
    return sc, rays


LR = [0.005, 0.0, 0.005, 0.005, 0.005, 0.0]


def _masks(sc):
    """rank-identical optimiser masks (the frustum mask of the current frame in the real Mapper); one level unmasked"""
    rng = np.random.default_rng(3)
    return {"middle": rng.random(sc["grids"]["middle"].shape[1:]) < 0.6, "fine": None, "color": rng.random(sc["grids"]["color"].shape[1:]) < 0.6}


def _worker(rank, world, port, out_path, kind="small", steps=2):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nice_slam_cpp_amd.dist as nd
    sc, rays = _scene_and_rays(kind)
    lo, hi = nd.shard_range(rays["rays_o"].shape[0], rank, world)
    be = OracleBackend(sc)
    be.o.lib.nso_set_num_threads(4 if kind == "k4" else 1)
    be.masks = _masks(sc)
    mapper = nd.ShardedMapper(be)
    sl = slice(lo, hi)
    gmaxes = []
    for _ in range(steps):
        g = mapper.step("color", rays["rays_o"][sl], rays["rays_d"][sl], torch.tensor(rays["gt_depth"][sl]).numpy() if False else rays["gt_depth"][sl],
                        rays["gt_color"][sl], LR, gt_depth_max=nd.global_depth_max(torch.tensor(rays["gt_depth"][sl])))
        gmaxes.append(g)
    np.savez(out_path % rank, fine=be.grids["fine"], color=be.grids["color"], dec=be.decs["color"], loss=be.loss, gmax=gmaxes[0], lo=lo, hi=hi)
    dist.destroy_process_group()


def test_two_rank_sharded_mapping_equals_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    import nice_slam_cpp_amd.dist as nd
    assert [nd.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    out = str(tmp_path / "rank%d.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    sc, rays = _scene_and_rays()
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 19, 19, 37)
    assert float(r0["gmax"]) == float(r1["gmax"]) == float(rays["gt_depth"].max())       # batch-global statistic
    # replicas stay bit-identical without any broadcast
    for k in ("fine", "color", "dec"):
        assert np.array_equal(r0[k], r1[k]), k
    assert float(r0["loss"]) == float(r1["loss"])
    # and equal the single-process full-batch result (fp64 oracle: only the summation order differs)
    be = OracleBackend(sc)
    be.masks = _masks(sc)
    m = nd.ShardedMapper(be)
    for _ in range(2):
        m.step("color", rays["rays_o"], rays["rays_d"], rays["gt_depth"], rays["gt_color"], LR, gt_depth_max=float(rays["gt_depth"].max()))
    for k, ref in (("fine", be.grids["fine"]), ("color", be.grids["color"]), ("dec", be.decs["color"])):
        assert np.abs(r0[k] - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), k
    assert abs(float(r0["loss"]) - be.loss) < 1e-9 * abs(be.loss)
    mk = _masks(sc)
    assert np.array_equal(r0["color"][:, ~mk["color"]], sc["grids"]["color"][:, ~mk["color"]])      # unmarked voxels: never sent, never moved


def test_two_rank_sharded_mapping_at_the_k4_shard_size(tmp_path):
    """the same at BASELINE configs[3]'s per-rank size: the office0-class grids (src/main.cpp:34-75 for that bound), 1250 rays x 48 per rank, rank-identical
    optimiser masks, one colour-stage step: the packed exchange (marked voxels + colour decoder + loss) and the replicated Adam step must leave both
    ranks bit-identical and equal to the single-process step on the 2500 rays"""
    sys.path.insert(0, ROOT)
    import nice_slam_cpp_amd.dist as nd
    out = str(tmp_path / "k4rank%d.npz")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out, "k4", 1), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    sc, rays = _scene_and_rays("k4")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 1250, 1250, 2500)
    assert float(r0["gmax"]) == float(r1["gmax"]) == float(rays["gt_depth"].max())
    for k in ("fine", "color", "dec"):
        assert np.array_equal(r0[k], r1[k]), k
    be = OracleBackend(sc)
    be.o.lib.nso_set_num_threads(8)
    be.masks = _masks(sc)
    m = nd.ShardedMapper(be)
    m.step("color", rays["rays_o"], rays["rays_d"], rays["gt_depth"], rays["gt_color"], LR, gt_depth_max=float(rays["gt_depth"].max()))
    for k, ref in (("fine", be.grids["fine"]), ("color", be.grids["color"]), ("dec", be.decs["color"])):
        assert np.abs(r0[k] - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), k
    assert abs(float(r0["loss"]) - be.loss) < 1e-9 * abs(be.loss)
    mk = _masks(sc)
    n_marked = int(mk["middle"].sum()) * 32 + sc["grids"]["fine"].size + int(mk["color"].sum()) * 32
    assert be.grad_pack().numel() == n_marked + sc["decoders"]["color"].size + 1                 # what travels: marked voxels of the three levels, decoder, loss


# ---- bundle adjustment over two ranks (BASELINE configs[4]) ---------------------------------------------------------------------------
BA_INTR = (40.0, 40.0, 32.0, 24.0)


def _ba_problem():
    """a three-frame window (the oldest pose fixed, two optimised), 13 rays per frame: 39 rays -> shards 20 + 19, the middle frame split"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes
    from oracle.nso import Oracle
    o = Oracle("f64")
    sc = scenes.make_scene(5, scenes.SMALL_GRID_SHAPES, grid_std=0.1)
    rng = np.random.default_rng(9)
    nf, per = 3, 13
    cams = np.zeros((nf, 8))
    for f in range(nf):
        q = np.array([1.0, 0.02 * f, -0.03 * f, 0.01 * f]); q /= np.linalg.norm(q)
        cams[f, :4] = q * (1.0 + 0.1 * f)                                # quad2rotation normalises: an unnormalised quaternion is a valid pose
        cams[f, 4:7] = np.array([-0.3 + 0.1 * f, 0.2, 0.1 - 0.05 * f])
    pix_i = rng.integers(4, 60, nf * per).astype(np.int32); pix_j = rng.integers(4, 44, nf * per).astype(np.int32)
    gt_depth = rng.uniform(0.8, 2.5, nf * per); gt_depth[5] = 0.0
    gt_color = rng.uniform(0, 1, (nf * per, 3))
    frames = [(f * per, per, f != 0) for f in range(nf)]
    return o, sc, cams, pix_i, pix_j, gt_depth, gt_color, frames


def _ba_rays(o, sc, cams, pix_i, pix_j, gt_depth, frames):
    ro, rd = np.zeros((pix_i.size, 3)), np.zeros((pix_i.size, 3))
    for f, (first, count, _) in enumerate(frames):
        c2w = o.camera_from_tensor(cams[f, :7])
        a, b = o.rays_from_pixels(pix_i[first:first + count], pix_j[first:first + count], *BA_INTR, c2w)
        ro[first:first + count], rd[first:first + count] = a, b
    keep = o.inside_filter(sc["bound"], ro, rd, gt_depth).astype(np.uint8)
    keep[7] = 0                                                           # a masked ray inside the first shard
    return ro, rd, keep


def _ba_run(group_world, steps=2):
    import nice_slam_cpp_amd.dist as nd
    o, sc, cams, pix_i, pix_j, gt_depth, gt_color, frames = _ba_problem()
    be = OracleBackend(sc)
    be.masks = _masks(sc)
    mapper = nd.ShardedMapper(be)
    nf = len(frames)
    cam_m, cam_v = np.zeros_like(cams), np.zeros_like(cams)
    xt = np.zeros(8 * nf + 8)
    N = pix_i.size
    g_ro, g_rd = np.zeros((N, 3)), np.zeros((N, 3))
    hist = []
    for it in range(steps):
        ro, rd, keep = _ba_rays(o, sc, cams, pix_i, pix_j, gt_depth, frames)       # the rays follow the poses: redrawn every iteration (Mapper.cpp:376-414)
        full = dict(rays_o=ro, rays_d=rd, gt_depth=gt_depth, gt_color=gt_color, pix_i=pix_i, pix_j=pix_j, keep=keep)
        g_ro[:] = 0; g_rd[:] = 0
        lo, hi = mapper.step_ba("color", full, frames, cams, cam_m, cam_v, BA_INTR, LR, 1e-3, it + 1, xt, g_ro, g_rd, w_color=0.2)
        hist.append((xt[:8 * nf].copy(), float(xt[8 * nf]), float(xt[8 * nf + 1]), lo, hi))
    return be, cams, hist, int(keep.sum())


def _ba_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be, cams, hist, nkeep = _ba_run(world)
    np.savez(out_path % rank, fine=be.grids["fine"], color=be.grids["color"], dec=be.decs["color"], cams=cams, g0=hist[0][0], g1=hist[1][0],
             loss=[h[1] for h in hist], kept=[h[2] for h in hist], lo=hist[0][3], hi=hist[0][4])
    dist.destroy_process_group()


def test_two_rank_bundle_adjustment_step_equals_single_process(tmp_path):
    """BASELINE configs[4] (joint pose + grid optimisation) sharded: two ranks run ShardedMapper.step_ba -- the shard's step with ray gradients,
    the shard's part of every window frame's pose gradient (nsk_pose_step_multi, step 0), ONE all-reduce of [marked voxels | colour decoder |
    loss | pose gradients | kept rays] (nsk_grad_extra), then the same Adam steps on grids, decoder and poses -- twice (the second iteration's
    rays follow the moved poses).  Against the single-process iterations on the whole batch: same pose gradients, losses, poses, grids and decoder
    (fp64 oracle: only the order of the sums differs); both ranks bit-identical; the shards' kept rays add up to the batch's; the window's fixed
    frame gets a zero gradient and does not move; the middle frame's rays are split between the ranks."""
    sys.path.insert(0, ROOT)
    out = str(tmp_path / "ba%d.npz")
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_ba_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 20, 20, 39)                # frame 1 (rays 13..25) lives on both ranks
    for k in ("fine", "color", "dec", "cams", "g0", "g1", "loss", "kept"):
        assert np.array_equal(r0[k], r1[k]), k
    be, cams, hist, nkeep = _ba_run(1)
    assert list(r0["kept"]) == [hist[0][2], hist[1][2]] and hist[0][2] == nkeep                          # shards add up to the batch
    for i, key in enumerate(("g0", "g1")):
        g = hist[i][0]
        assert np.abs(g[:8]).max() == 0 and np.abs(r0[key][:8]).max() == 0                               # the oldest frame is not optimised (Mapper.cpp:305-329)
        assert np.abs(g[8:]).max() > 0
        assert np.abs(r0[key] - g).max() < 1e-9 * np.abs(g).max(), key
    assert np.allclose(r0["loss"], [h[1] for h in hist], rtol=1e-12)
    assert np.abs(r0["cams"] - cams).max() < 1e-10 and np.abs(cams[0, :7] - _ba_problem()[2][0, :7]).max() == 0
    assert np.abs(cams[1:, :7] - _ba_problem()[2][1:, :7]).max() > 1e-4                                   # two Adam steps of 1e-3 moved the optimised poses
    for k, ref in (("fine", be.grids["fine"]), ("color", be.grids["color"]), ("dec", be.decs["color"])):
        assert np.abs(r0[k] - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), k
