#!/usr/bin/env python3
"""Benchmark of the hot path: one mapping iteration (cell sort + render forward + Mapper loss + backward + [all-reduce] +
Adam) per step.  Prints ONE JSON line (rank 0).

Headline workload (N = 1): BASELINE.json configs[2], the largest single-GPU configuration -- a ScanNet-scene0000_00-class
room (bound and intrinsics declared in tests/scenes.py: the reference has no such config), 5000 rays x 48 samples, colour-stage
iteration (renders the middle + fine + colour decoders, i.e. a superset of the fine stage; the fine-stage iteration of the same
workload, configs[1] (config/nice_slam.yaml, 1000 rays) and a configs[3] shard (1250 rays) are reported under "extras").

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rays shard across ranks with no data-path exchange except ONE all-reduce (RCCL, sum, fp32) of the gradient slab per step
(SURVEY.md 8e); per-GPU work is fixed, so scaling is "weak".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_16BIT_MFMA_TFLOPS = 2500.0    # same guide, dense BF16 / FP16 MFMA peak
PEAK_HBM_GBS = 8000.0              # same guide, HBM3E peak
# committed rocprofv3 counter passes of this same (default) command, newest first (tools/profile_round.sh): FETCH_SIZE / WRITE_SIZE and the
# SQ passes (instruction counts, MFMA busy cycles, waits ...) + GRBM_GUI_ACTIVE.  bench.py cannot collect counters itself; what it replays from
# these files is labelled "counters_from" in the JSON line -- the durations beside them are always this run's HIP events.
PMC_FILES = ("r04_pmc_hbm.json", "r03_pmc_hbm.json")
PMC_SQ_FILES = ("r04_pmc_sq.json", "r03_pmc_sq.json")
KERNEL_PREFIX = {"decode_bwd_multi": "void k_decode_bwd_multi<false>", "decode_fwd_multi": "void k_decode_fwd_multi_"}      # (_occ<8>: the merged middle + fine role, round 4; _bf16<8, 2> before)


def _first_profile(names):
    for n in names:
        if os.path.exists(os.path.join(ROOT, "profiles", n)):
            return n
    return None

# algorithmic MACs per sample of each decoder role (DESIGN.md "Kernels")
MAC = {
    "fwd_coarse": 6176, "fwd_middle": 15479, "fwd_fine": 20599, "fwd_color": 15575,
    "bwd_coarse": 6176, "bwd_middle": 9248, "bwd_fine": 9248, "bwd_color": 9344,
    # trainable: input gradients + embedding-gradient products + weight gradients (block outputs are saved by the forward)
    "bwd_color_train": 9344 + 5952 + 15575,
}
# 16-bit matrix products the kernels spend per fp32 multiply-add of the table above: two fp16 pieces = 3 products
# (forward and backward chains), two bf16 pieces = 4 products (weight-gradient panels); the coarse decoder runs on the fp32 MFMA
PRODUCTS = {
    "fwd_coarse": 0.0, "fwd_middle": 3.0, "fwd_fine": 3.0, "fwd_color": 3.0,
    "bwd_coarse": 0.0, "bwd_middle": 3.0, "bwd_fine": 3.0, "bwd_color": 3.0,
    "bwd_color_train": (3.0 * (9344 + 5952) + 4.0 * 15575) / (9344 + 5952 + 15575),
}
# ALGORITHMIC HBM bytes per sample, exactly SURVEY.md 8(d): 8 corners x 32 ch x 4 B = 1024 B per level looked up, a scatter is a
# read-modify-write (2048 B per level that receives gradient), ray I/O 5 B per sample; Adam 28 B per MARKED parameter per step
ALG_BYTES_FWD = {"coarse": 1024 + 5, "middle": 1024 + 5, "fine": 2048 + 5, "color": 3072 + 5}
ALG_BYTES_BWD = {"coarse": 2048, "middle": 2048, "fine": 2 * 2048, "color": 3 * 2048}
ADAM_BYTES_PER_PARAM = 28.0
# what the IMPLEMENTATION moves on top of that, per sample (reported as impl_bytes, never priced as algorithmic): the backward re-reads the
# features of a trainable decoder's level (1024), per-sample intermediates (z, outputs, g_raw, 32 B of ReLU bits per decoder) and the
# trainable decoder's saved block outputs (640 B written by the forward, read by the backward)
IMPL_BYTES = {
    "fwd_coarse": 1024 + 8, "fwd_middle": 1024 + 8, "fwd_fine": 2048 + 8, "fwd_color": 1024 + 20 + 640,
    "bwd_coarse": 2048 + 20, "bwd_middle": 2048 + 52, "bwd_fine": 2048 + 52, "bwd_color": 2048 + 52,
    "bwd_color_train": 3072 + 20 + 640,
}
STAGE_DECODERS = {"coarse": ["coarse"], "middle": ["middle"], "fine": ["middle", "fine"], "color": ["middle", "fine", "color"]}
# mapping.stage.<stage> learning rates of config/nice_slam.yaml:71-95 (groups: decoders, coarse, middle, fine, color, camera)
STAGE_LR = {"coarse": [0.0, 0.001, 0.0, 0.0, 0.0, 0.0], "middle": [0.0, 0.0, 0.1, 0.0, 0.0, 0.0],
            "fine": [0.0, 0.0, 0.005, 0.005, 0.0, 0.0], "color": [0.005, 0.0, 0.005, 0.005, 0.005, 0.0]}


def workloads():
    import scenes
    return {
        "K3": dict(bound=scenes.K3_BOUND, cam=scenes.CAM_SCANNET, rays=5000, up="z",
                   name="configs[2]: ScanNet scene0000_00-class room (bound [[0,8.6],[0,8.9],[-0.3,3.3]], 640x480 camera; declared in the "
                        "harness, the reference holds no such config), grids of src/main.cpp:34-75 for that bound"),
        "K2": dict(bound=scenes.REF_BOUND, cam=scenes.CAM_NICE_SLAM, rays=1000, up="y",
                   name="configs[1]: config/nice_slam.yaml grids (bound of src/main.cpp:33, Replica-room0-class), 1200x680 camera"),
        "K4": dict(bound=scenes.K4_BOUND, cam=scenes.CAM_NICE_SLAM, rays=1250, up="z",
                   name="configs[3] shard: Replica office0-class room, 10000 rays / 8 GPUs = 1250 rays per GPU"),
    }


def alg_counts(stage, trainable_color):
    """per sample, for the forward and the backward launch of a stage: MACs, SURVEY 8(d) bytes, implementation bytes, 16-bit products per MAC"""
    decs = STAGE_DECODERS[stage]
    bk = lambda d: "bwd_color_train" if (d == "color" and trainable_color) else "bwd_" + d
    fm = sum(MAC["fwd_" + d] for d in decs)
    bm = sum(MAC[bk(d)] for d in decs)
    fp = sum(MAC["fwd_" + d] * PRODUCTS["fwd_" + d] for d in decs) / max(fm, 1)
    bp = sum(MAC[bk(d)] * PRODUCTS[bk(d)] for d in decs) / max(bm, 1)
    return {"fwd": dict(mac=fm, alg_bytes=ALG_BYTES_FWD[stage], impl_bytes=sum(IMPL_BYTES["fwd_" + d] for d in decs), products=fp),
            "bwd": dict(mac=bm, alg_bytes=ALG_BYTES_BWD[stage], impl_bytes=sum(IMPL_BYTES[bk(d)] for d in decs), products=bp)}


def cpu_threads():
    """host threads a CPU baseline may use: the process's CPU share (cgroup quota if any, else its affinity mask), capped at the
    GPU box's documented per-GPU share of 16 cores (NSK_CPU_THREADS overrides) -- the box shows 256 logical CPUs, and 256
    threads on these small ops ran the ATen step 600x slower than 16 do"""
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("NSK_CPU_THREADS", "16"))))


def cpu_baseline_c(sc, rays_list, stage, lr, w_color, seconds):
    """the oracle (a plain-C port of the reference's path, oracle/nso.c, OpenMP over rays) timed on this host: same step, same inputs"""
    from oracle.nso import Oracle
    o = Oracle("f32")
    o.lib.nso_set_num_threads(cpu_threads())
    threads = int(o.lib.nso_num_threads())
    grids = {k: v.copy() for k, v in sc["grids"].items()}
    decs = {k: v.copy() for k, v in sc["decoders"].items()}
    levels = [k for k in STAGE_DECODERS[stage]]
    mom = {k: (np.zeros_like(grids[k]), np.zeros_like(grids[k])) for k in levels}
    dm, dv = np.zeros_like(decs["color"]), np.zeros_like(decs["color"])
    op = o.opts(sc["bound"])
    n_rays, steps, t0 = 0, 0, time.perf_counter()
    while True:
        r = rays_list[steps % len(rays_list)]
        fw = o.render_forward(op, grids, decs, stage, r["rays_o"], r["rays_d"], r["gt_depth"])
        _, g_d, g_c = o.loss_map(fw["depth"], fw["rgb"], r["gt_depth"], r["gt_color"], w_color, stage == "color")
        bw = o.render_backward(op, grids, decs, stage, r["rays_o"], r["rays_d"], r["gt_depth"], -1.0, g_c, g_d, None, want_rays=False)
        steps += 1
        for k in mom:
            o.adam_step(grids[k], bw["g_grids"][k], mom[k][0], mom[k][1], lr[2], steps)
        if stage == "color":
            o.adam_step(decs["color"], bw["g_decoders"]["color"], dm, dv, lr[0], steps)
        n_rays += r["rays_o"].shape[0]
        if time.perf_counter() - t0 > seconds or steps >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": n_rays / dt, "unit": "rays/s", "cores": threads, "kind": "port", "impl": "oracle/nso.c (plain C, analytic backward, OpenMP over rays)",
            "sample": "%d full mapping steps of %d rays x 48 samples (%s stage, same scene), %.1f s" % (steps, rays_list[0]["rays_o"].shape[0], stage, dt),
            "ms_per_step": 1e3 * dt / steps}


def cpu_baseline_aten(sc, rays_list, stage, lr, w_color, seconds):
    """the reference's execution style on the host: the ATen op sequence the reference calls (F.grid_sample, F.linear, sort, cumprod,
    autograd, torch.optim.Adam -- oracle/torch_ref.py restates src/Renderer.cpp, src/models/*.cpp, utils.h, src/Mapper.cpp:330-446 op for
    op, SURVEY.md section 0.3 semantics) with torch.set_num_threads(all host cores of this process)"""
    from oracle import torch_ref as T
    threads = cpu_threads()
    torch.set_num_threads(threads)
    bound = torch.tensor(np.asarray(sc["bound"], np.float32))
    levels = STAGE_DECODERS[stage]
    grids = {k: torch.tensor(v[None].copy()) for k, v in sc["grids"].items()}
    decs = {k: torch.tensor(v.copy()) for k, v in sc["decoders"].items()}
    groups = []
    for gi, k in ((2, "middle"), (3, "fine"), (4, "color"), (1, "coarse")):
        if k in levels:
            grids[k].requires_grad_(True)
            groups.append({"params": [grids[k]], "lr": lr[gi]})
    if stage == "color":
        decs["color"].requires_grad_(True)
        groups.append({"params": [decs["color"]], "lr": lr[0]})
    opt = torch.optim.Adam(groups)                                              # src/Mapper.cpp:330
    tens = [{k: torch.tensor(r[k]) for k in ("rays_o", "rays_d", "gt_depth", "gt_color")} for r in rays_list]

    def step(i):
        r = tens[i % len(tens)]
        opt.zero_grad()
        rgb, depth, var, w = T.render_batch_ray(grids, decs, r["rays_d"], r["rays_o"], stage, r["gt_depth"], bound)
        loss = T.loss_map(depth, rgb, r["gt_depth"], r["gt_color"], w_color, stage == "color")
        loss.backward()
        opt.step()
        return r["rays_o"].shape[0]

    step(0)                                                                     # warm-up (thread pool, allocator)
    n_rays, steps, t0 = 0, 0, time.perf_counter()
    while True:
        n_rays += step(steps + 1)
        steps += 1
        if time.perf_counter() - t0 > seconds or steps >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": n_rays / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "impl": "ATen-CPU op sequence of the reference with autograd + torch.optim.Adam (oracle/torch_ref.py), torch %s, %d threads" % (torch.__version__, threads),
            "sample": "%d full mapping steps of %d rays x 48 samples (%s stage, same scene and rays as the GPU run), %.1f s after 1 warm-up step"
                      % (steps, rays_list[0]["rays_o"].shape[0], stage, dt),
            "ms_per_step": 1e3 * dt / steps}


TUNE = []


def run_workload(wl, stage, N, steps, warmup, local, rank, world, dist, graph=False, frustum=True, matmul_mode=None, pipeline=False,
                 comm=None, rays_total=None, repeats=1, backward_mode=None, forward_only=False):
    """time `steps` mapping iterations of workload `wl` at `N` rays per GPU (rays_total: a FIXED batch of that many rays sharded over
    the ranks instead -- strong scaling); returns dict(dt, prof, loss, scene, pool, ...).  comm: an RCCL communicator for the C-ABI
    exchange (nsk_allreduce_grads), None = torch.distributed on the packed buffer"""
    import nice_slam_cpp_amd as pkg
    import nice_slam_cpp_amd.dist as nd
    import scenes
    dev = torch.device("cuda", local)
    cam = wl["cam"]
    sc = scenes.make_scene(42, scenes.grid_shapes_for(wl["bound"]), bound=wl["bound"])     # grid shapes + init of src/main.cpp:33-78
    # one optimize_map call: a fixed window of 5 frames (4 keyframes + the current frame, mapping_window_size 5), fresh random pixels of
    # those frames every iteration (src/Mapper.cpp:376-414); every rank draws its own pixels of the same frames
    if rays_total is None:
        pool = [scenes.make_rays(1234 + 17 * i + 1000 * rank, N, sc["bound"], n_frames=5, cam_seed=4242, up=wl["up"], **cam) for i in range(8)]
    else:                                                        # the same batch on every rank, each takes its contiguous shard
        lo, hi = nd.shard_range(rays_total, rank, world)
        N = hi - lo
        pool = []
        for i in range(8):
            full = scenes.make_rays(1234 + 17 * i, rays_total, sc["bound"], n_frames=5, cam_seed=4242, up=wl["up"], **cam)
            pool.append({k: (v[lo:hi] if isinstance(v, np.ndarray) and v.shape[:1] == (rays_total,) else v) for k, v in full.items()})
    ctx = pkg.Context(local)
    for kv in TUNE:                                  # --tune key=value: nsk_set_tuning experiments (include/nsk.h)
        k, v = kv.split("=")
        if k == "sort_mode":                             # (nsk_set_sort_mode: -1 automatic, 0 ray order, 1 cell-sorted)
            ctx.set_sort_mode(int(v))
        else:
            ctx.set_tuning(k, int(v))
    ctx.set_render_opts()                                        # 32 + 16 samples (src/Renderer.cpp:9-10)
    if matmul_mode is not None:
        ctx.set_matmul_mode(matmul_mode)
    if backward_mode is not None:
        ctx.set_backward_mode(backward_mode)
    ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"])
    mask_frac = None
    if frustum:                                                  # mapping.frustum_feature_selection: True (nice_slam.yaml:62): the optimiser
        c2w_cur = pool[0]["c2w"][-1]                             # parameters are the voxels in the current frame's frustum (Mapper.cpp:254-290)
        depth_img = torch.tensor(scenes.frame_depth_image(sc["bound"], c2w_cur, **cam), device=dev)
        intr = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        mask_frac = {k: float(ctx.frustum_mask(k, depth_img, intr, c2w_cur).mean()) for k in ("coarse", "middle", "fine", "color")}
    train_color = stage == "color"
    ctx.decoder_set_trainable("color", train_color)              # fix_fine: True, fix_color: False (nice_slam.yaml:51-52)
    lr = STAGE_LR[stage]
    w_color = 0.5                                                # src/Mapper.cpp:33 reads tracking.w_color_loss (D20)
    cu = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev).contiguous()
    batches = []
    for r in pool:
        gd = cu(r["gt_depth"])
        gmax = nd.global_depth_max(gd)                           # batch-global max(gt_depth) (Renderer.cpp:76,93)
        batches.append((cu(r["rays_o"]), cu(r["rays_d"]), gd, cu(r["gt_color"]), gmax))
    loss = torch.zeros(1, device=dev)
    flags = pkg.nsk.GRAD_GRIDS | (pkg.nsk.GRAD_DECODERS if train_color else 0)
    xev = []                                                     # (start, stop) events around the exchange when it goes through torch.distributed

    with torch.cuda.stream(ctx.tstream):
        xn = [0]
        timing = [False]

        def step(i):
            ro, rd, gd, gc, gmax = batches[i % len(batches)]
            if forward_only:                                     # SURVEY 8(d)'s second figure: Renderer::render_batch_ray alone (sampling, decoders, compositing)
                ctx.render_forward(stage, ro, rd, gd, gmax, want_weights=False)
                return
            if pipeline and not graph:                           # the next batch is registered first: its sampling + cell sort ride in this step's
                nro, nrd, ngd, _, ngmax = batches[(i + 1) % len(batches)]      # composite / backward / Adam launches (nsk_map_prepare)
                ctx.map_prepare(stage, nro, nrd, ngd, ngmax, flags=flags)
            ctx.map_step(stage, ro, rd, gd, gc, gmax, w_color, stage == "color", flags=flags, loss=loss)
            if world > 1:                                        # the one exchange of the path: the marked voxels of the touched levels,
                if comm is not None:                             # the colour decoder's gradient and the loss
                    ctx.allreduce_grads_rccl(comm)               # pack -> ncclAllReduce -> unpack on the context's stream, no Python between them
                else:
                    buf = ctx.grad_pack()
                    xn[0] = buf.numel()
                    if timing[0]:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); nd.allreduce_grads(buf); e1.record(); xev.append((e0, e1))
                    else:
                        nd.allreduce_grads(buf)
                    ctx.grad_unpack()
            ctx.adam_step(lr)

        eager_step = step
        if graph and world == 1:                                 # one graph per batch of the pool, replayed instead of re-issued
            step(0)                                              # sizes the workspaces
            gids = []
            for b in range(len(batches)):
                ctx.graph_begin(); step(b); gids.append(ctx.graph_end())

            def step(i):                                         # noqa: F811
                ctx.graph_launch(gids[i % len(gids)])
        for i in range(warmup):
            step(i)
        # The timed region is EXACTLY `steps` steps between barrier + synchronize on both sides, maximum over the ranks -- and that block is
        # run `repeats` times back to back, the MEDIAN block being the one reported: a 20-step block of a 0.4 ms step is 8 ms behind a few
        # warm-up steps, clocks and caches are not settled, and the driver's 20-step line read 7-10 % below a 200-step run of the same box.
        blocks = []
        for rep in range(max(1, repeats)):
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                step(warmup + rep * steps + i)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            blocks.append(time.perf_counter() - t0)
        if dist is not None:
            t = torch.tensor(blocks, device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            blocks = [float(x) for x in t]
        dt = float(np.median(blocks))
        # per-kernel durations: HIP events recorded on the context's stream around every launch (same steps again)
        ctx.profile_begin()
        timing[0] = True
        for i in range(steps):
            eager_step(warmup + i)
        timing[0] = False
        prof = ctx.profile_end()
        if xev:
            torch.cuda.synchronize()
            prof["allreduce"] = (len(xev), sum(a.elapsed_time(b) for a, b in xev))
        final_loss = float(loss)
        # what an exchange carries (reported also for N = 1)
        ctx.map_step(stage, *batches[0][:4], batches[0][4], w_color, stage == "color", flags=flags, loss=loss)
        xn[0] = ctx.grad_pack().numel()
        ctx.zero_grads()
    marked = None
    if mask_frac is not None:
        marked = sum(mask_frac[k] * sc["grids"][k].size for k in STAGE_DECODERS[stage])
    out = dict(dt=dt, blocks=blocks, prof=prof, loss=final_loss, sc=sc, pool=pool, lr=lr, w_color=w_color, slab_floats=int(ctx.grad_slab().numel()),
               exchange_floats=int(xn[0]), mask_frac=mask_frac, marked_params=marked, rays_per_gpu=N)
    ctx.close()
    return out


def quat_cam(c2w):
    """c2w [4,4] -> pose 7-vector (qw, qx, qy, qz, tx, ty, tz) padded to 8 floats (get_tensor_from_camera, utils.h:212-231)"""
    R = c2w[:3, :3].astype(np.float64)
    qw = np.sqrt(max(1e-12, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    q = np.array([qw, (R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw)])
    return np.concatenate([q, c2w[:3, 3], [0.0]]).astype(np.float32)


def run_k5_loop(local, rank, world, dist, comm, cycles=20, warmup=3, track_rays=200, track_iters=10, map_rays=1000, map_iters=12, tune=()):
    """BASELINE configs[4] as a loop: per frame of a TUM-fr1/desk-class sequence (bound / camera declared in tests/scenes.py), the Tracker's
    iterations on 200 pixels of the new frame (config/nice_slam.yaml tracking.pixels / iters; on rank 0 only, its pose then goes to every rank: 8
    floats), then the frame's share of the Mapper's iterations (mapping.iters 60 on every 5th frame = 12 per frame) on 1000 pixels of a
    5-frame window in the colour stage WITH bundle adjustment (4 poses optimised, the oldest fixed: src/Mapper.cpp:305-329): rays sharded over
    the ranks, ONE all-reduce per iteration of [marked voxels | colour decoder | loss | 40 floats of pose gradient], replicated Adam on grids,
    decoder and poses (ShardedMapper.step_ba).  Everything an iteration needs is drawn on the device from the current poses (nsk_prepare_rays)."""
    import nice_slam_cpp_amd as pkg
    import nice_slam_cpp_amd.dist as nd
    import scenes
    dev = torch.device("cuda", local)
    cam = scenes.CAM_TUM
    intr = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    H, W = cam["H"], cam["W"]
    sc = scenes.make_scene(42, scenes.grid_shapes_for(scenes.K5_BOUND), bound=scenes.K5_BOUND)
    rng = np.random.default_rng(777)
    nf = 5
    c2ws = [scenes.make_camera(rng, sc["bound"], "z") for _ in range(nf)]
    cu = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device=dev).contiguous()
    depth = [cu(scenes.frame_depth_image(sc["bound"], c, **cam)) for c in c2ws]
    color = [cu(scenes.frame_color_image(sc["bound"], c, **cam)) for c in c2ws]
    ctx = pkg.Context(local)
    for kv in tune:                                        # experiments: nsk_set_tuning keys (--tune key=value)
        k_, v_ = kv.split("="); ctx.set_tuning(k_, int(v_))
    for kv in TUNE:
        k, v = kv.split("=")
        ctx.set_sort_mode(int(v)) if k == "sort_mode" else ctx.set_tuning(k, int(v))
    ctx.set_render_opts()
    ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"])
    for k in ("coarse", "middle", "fine", "color"):
        ctx.frustum_mask(k, depth[-1], intr, c2ws[-1])
    ctx.decoder_set_trainable("color", True)
    cams = cu(np.stack([quat_cam(c) for c in c2ws]))                 # [nf, 8], replicated
    cam_m, cam_v = torch.zeros_like(cams), torch.zeros_like(cams)
    trk = cu(quat_cam(c2ws[-1])); trk_m, trk_v = torch.zeros(8, device=dev), torch.zeros(8, device=dev)
    per = map_rays // nf
    N = per * nf
    frames = [(f * per, per, f != 0) for f in range(nf)]
    xt = torch.zeros(8 * nf + 8, device=dev)
    g_ro, g_rd = torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev)
    tg_ro, tg_rd = torch.zeros(track_rays, 3, device=dev), torch.zeros(track_rays, 3, device=dev)
    loss = torch.zeros(1, device=dev)
    mapper = nd.ShardedMapper(ctx, comm=comm)
    lr = STAGE_LR["color"]
    edge = 20                                                        # tracking.ignore_edge_W / _H
    seed = [1000]
    ba_step, trk_step = [0], [0]

    def cycle():
        if rank == 0:                                                # ---- Tracker (src/Tracker.cpp:92-113) on the newest frame
            for _ in range(track_iters):
                seed[0] += 1
                r = ctx.prepare_rays([dict(depth=depth[-1], color=color[-1], pose=trk[:7], seed=seed[0])], track_rays, (edge, H - edge, edge, W - edge), intr)
                ctx.set_ray_mask(r["keep"])
                ctx.track_step("color", r["rays_o"], r["rays_d"], r["gt_depth"], r["gt_color"], -1.0, 0.5, True, True, True, flags=4, loss=loss, g_rays=(tg_ro, tg_rd))
                ctx.set_ray_mask(None)
                trk_step[0] += 1
                ctx.pose_step(r["pix_i"], r["pix_j"], intr, tg_ro, tg_rd, trk, trk_m, trk_v, 1e-3, trk_step[0])
        if world > 1:                                                # the tracked pose to every rank (28 bytes of payload)
            dist.broadcast(trk, src=0)
        cams[nf - 1].copy_(trk)
        for _ in range(map_iters):                                   # ---- Mapper, colour stage with bundle adjustment (src/Mapper.cpp:366-368,430-446)
            seed[0] += 1
            full = ctx.prepare_rays([dict(depth=depth[f], color=color[f], pose=cams[f, :7], seed=seed[0] * 131 + f) for f in range(nf)], per, (0, H, 0, W), intr)
            ba_step[0] += 1
            mapper.step_ba("color", full, frames, cams, cam_m, cam_v, intr, lr, 1e-3, ba_step[0], xt, g_ro, g_rd, w_color=0.5)

    with torch.cuda.stream(ctx.tstream):
        for _ in range(warmup):
            cycle()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(cycles):
            cycle()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        ctx.profile_begin()
        cycle()
        prof = ctx.profile_end()
    ctx.sync()
    out = {"workload": "configs[4]: TUM-fr1/desk-class volume (bound [[-3.5,3],[-3,3],[-3,3]], 640x480 camera; declared in the harness), per frame: Tracker %d iterations x %d "
                       "pixels (rank 0) + pose to all ranks + Mapper %d colour-stage iterations x %d pixels over a %d-frame window with bundle adjustment (4 poses), rays "
                       "sharded x%d" % (track_iters, track_rays, map_iters, N, nf, world),
           "ms_per_frame": 1e3 * dt / cycles, "frames": cycles,
           "value": map_iters * N * cycles / dt, "unit": "mapping rays/s (whole job)", "mapping_rays_per_frame": map_iters * N, "tracking_rays_per_frame": track_iters * track_rays,
           "exchange_floats_ba": int(xt.numel()), "final_ba_loss": float(xt[8 * nf]), "kept_rays_last_iteration": float(xt[8 * nf + 1]),
           "kernels_us_per_frame": {k: round(1e3 * ms, 1) for k, (c, ms) in prof.items()}, "launches_per_frame": int(sum(c for c, ms in prof.values())),
           "kernel_ms_per_frame_sum": 1e-0 * sum(ms for c, ms in prof.values()),
           "note": "ms_per_frame is the wall time of this Python loop (about ten ctypes calls per iteration: host-bound on a slow host); kernel_ms_per_frame_sum is the "
                   "HIP-event sum of the profiled kernels of one frame.  The C++ Mapper / Tracker are the product hosts (tests/test_gpu_host_cpp.py prints their loop times).",
           "hardware": "unmeasured on a multi-GPU node unless n_gpus > 1 in this line" if world == 1 else "%d ranks" % world}
    ctx.close()
    return out


def sq_profile(kernel_prefix):
    """MFMA-busy share of the dominant kernel from the committed SQ counter passes of this same command (profiles/, tools/profile_round.sh):
    SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x the kernel's cycles); the kernel's cycles = GRBM_GUI_ACTIVE / 8 XCDs where that
    counter was collected, else duration x 2.4 GHz (an upper bound on the cycles, so a lower bound on the share)"""
    name = _first_profile(PMC_SQ_FILES)
    if name is None:
        return None
    pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
    get = lambda c: next((v for k, v in pmc.items() if k.startswith(kernel_prefix) and k.endswith("|" + c)), None)
    busy, gui = get("SQ_VALU_MFMA_BUSY_CYCLES"), get("GRBM_GUI_ACTIVE")
    if busy is None:
        return None
    return {"SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui, "SQ_WAIT_ANY": get("SQ_WAIT_ANY"), "SQ_WAVE_CYCLES": get("SQ_WAVE_CYCLES"),
            "SQ_INSTS_VALU": get("SQ_INSTS_VALU"), "SQ_INSTS_MFMA": get("SQ_INSTS_MFMA"), "SQ_INSTS_VMEM": get("SQ_INSTS_VMEM"), "SQ_INSTS_LDS": get("SQ_INSTS_LDS"),
            "SQ_LDS_BANK_CONFLICT": get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": get("SQ_LDS_IDX_ACTIVE"), "file": "profiles/" + name}


def counter_fracs(sq, dur_s):
    """what bounds a decoder kernel, from its counters: SIMD-cycles = 4 SIMDs x 256 CUs x the kernel's cycles (GRBM_GUI_ACTIVE / 8 XCDs, else
    duration x 2.4 GHz); mfma_busy = matrix-pipe busy cycles / SIMD-cycles; issue_frac = (4 cycles per non-MFMA vector instruction + the
    matrix pipe's busy cycles) / SIMD-cycles -- matrix and vector instructions of a SIMD serialise (tools/ubench/interleave.hip);
    wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of a wave's resident cycles spent waiting on a counter)"""
    cycles = sq["GRBM_GUI_ACTIVE"] / 8.0 if sq.get("GRBM_GUI_ACTIVE") else dur_s * 2.4e9
    simd = 4.0 * 256.0 * cycles
    out = {"simd_cycles": simd, "mfma_busy": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / simd}
    if sq.get("SQ_INSTS_VALU") is not None and sq.get("SQ_INSTS_MFMA") is not None:
        out["valu_issue_frac"] = 4.0 * (sq["SQ_INSTS_VALU"] - sq["SQ_INSTS_MFMA"]) / simd
        out["issue_frac"] = out["valu_issue_frac"] + out["mfma_busy"]
    if sq.get("SQ_WAIT_ANY") is not None and sq.get("SQ_WAVE_CYCLES"):
        out["wait_frac"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
    if sq.get("SQ_LDS_BANK_CONFLICT") is not None and sq.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_conflict_frac"] = sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]
    return out


def summarize(res, stage, N, steps, world, with_counters=False):
    S = 48
    M = N * S
    prof = res["prof"]
    per_kernel = {k: {"launches": c, "avg_us": 1e3 * ms / c} for k, (c, ms) in prof.items()}
    cnt = alg_counts(stage, stage == "color")
    fwd_name = "decode_fwd_multi" if "decode_fwd_multi" in prof else next(k for k in prof if k.startswith("decode_fwd"))
    bwd_name = "decode_bwd_multi" if "decode_bwd_multi" in prof else next(k for k in prof if k.startswith("decode_bwd"))
    alg = {fwd_name: cnt["fwd"], bwd_name: cnt["bwd"]}
    dom = max(alg, key=lambda k: prof[k][1])
    dom_s = prof[dom][1] / prof[dom][0] * 1e-3
    flops = 2.0 * alg[dom]["mac"] * M
    tf = flops / dom_s / 1e12
    products = alg[dom]["products"]
    # "bound": the contract's vocabulary is hbm | mfma, and this launch is neither: its counters (committed passes, replayed below under
    # "counters") say vector-instruction issue + latency -- `frac` stays the algorithmic fp32 FLOP rate over the fp32 matrix peak because the
    # contract is fp32 arithmetic, but that peak is not what limits a chain run on 16-bit pieces (the forward exceeds it), so read
    # issue_frac / wait_frac / mfma_busy / frac_16bit beside it
    roof = {"bound": "valu-issue+latency", "contract_bound": "mfma", "kernel": dom, "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
            "avg_launch_us": dom_s * 1e6, "alg_flops_per_launch": flops,
            # the same launch against what its own instruction mix allows: every fp32 multiply-add is `products` 16-bit matrix products
            "products_per_mac": products, "frac_16bit": (tf * products / PEAK_16BIT_MFMA_TFLOPS) if products > 0 else None,
            "peak_16bit": PEAK_16BIT_MFMA_TFLOPS,
            # HBM view of the same launch: SURVEY 8(d) bytes only; what the implementation moves on top is impl_bytes_per_launch
            "alg_bytes_per_launch": float(alg[dom]["alg_bytes"]) * M, "hbm_frac_same_kernel": alg[dom]["alg_bytes"] * M / dom_s / 1e9 / PEAK_HBM_GBS,
            "impl_bytes_per_launch": float(alg[dom]["impl_bytes"]) * M, "mfma_busy": None}
    if with_counters:
        sq = sq_profile(KERNEL_PREFIX[dom]) if dom in KERNEL_PREFIX else None
        if sq:
            roof.update({k: v for k, v in counter_fracs(sq, dom_s).items() if k != "simd_cycles"})
            roof["counters"] = sq
            roof["counters_from"] = sq["file"] + " (committed rocprofv3 --pmc passes of this command, replayed: NOT measured in this run)"
        other = fwd_name if dom == bwd_name else bwd_name      # the sibling launch on the same denominators
        sq2 = sq_profile(KERNEL_PREFIX[other]) if other in KERNEL_PREFIX else None
        if sq2:
            o_s = prof[other][1] / prof[other][0] * 1e-3
            o_tf = 2.0 * alg[other]["mac"] * M / o_s / 1e12
            roof["sibling"] = dict({"kernel": other, "avg_launch_us": o_s * 1e6, "achieved": o_tf, "frac": o_tf / PEAK_FP32_MFMA_TFLOPS,
                                    "frac_16bit": o_tf * alg[other]["products"] / PEAK_16BIT_MFMA_TFLOPS},
                                   **{k: v for k, v in counter_fracs(sq2, o_s).items() if k != "simd_cycles"})
    step_s = res["dt"] / steps
    nparam = sum(res["sc"]["grids"][k].size for k in STAGE_DECODERS[stage])
    marked = res["marked_params"] if res.get("marked_params") is not None else nparam
    if stage == "color":
        marked += res["sc"]["decoders"]["color"].size
    alg_step = float(cnt["fwd"]["alg_bytes"] + cnt["bwd"]["alg_bytes"]) * M + ADAM_BYTES_PER_PARAM * marked
    impl_step = float(cnt["fwd"]["impl_bytes"] + cnt["bwd"]["impl_bytes"]) * M + ADAM_BYTES_PER_PARAM * marked
    step_flops = 2.0 * (cnt["fwd"]["mac"] + cnt["bwd"]["mac"]) * M + 1000.0 * M     # + sampling / compositing
    blocks = res.get("blocks") or [res["dt"]]
    return {"value": world * N / step_s, "ms_per_step": 1e3 * step_s, "repeats": len(blocks),
            "block_ms_per_step": {"median": 1e3 * step_s, "min": 1e3 * min(blocks) / steps, "max": 1e3 * max(blocks) / steps, "first": 1e3 * blocks[0] / steps},
            "roofline": roof,
            "step_rooflines": {"alg_bytes_per_step": alg_step, "hbm_frac": alg_step / step_s / 1e9 / PEAK_HBM_GBS,
                               "impl_bytes_per_step": impl_step, "adam_marked_params": marked,
                               "alg_flops_per_step": step_flops, "fp32_frac": step_flops / step_s / 1e12 / PEAK_FP32_MFMA_TFLOPS},
            "kernels": per_kernel, "final_loss": res["loss"]}


def extra_line(name, wl, stage, r, s, steps):
    x = {"workload": wl["name"], "stage": stage, "rays_per_gpu": r["rays_per_gpu"], "steps": steps, "value": s["value"], "unit": "rays/s",
         "ms_per_step": s["ms_per_step"], "roofline_frac": s["roofline"]["frac"], "roofline_kernel": s["roofline"]["kernel"],
         "hbm_frac_step": s["step_rooflines"]["hbm_frac"], "kernels_avg_us": {kk: round(v["avg_us"], 2) for kk, v in s["kernels"].items()}}
    if "allreduce" in s["kernels"]:
        x["allreduce_us"] = s["kernels"]["allreduce"]["avg_us"]
        x["exchange_bytes"] = 4 * r["exchange_floats"]
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=9, help="the timed block of --steps steps is run this many times back to back; the median block is reported")
    ap.add_argument("--workload", default="K3", choices=["K3", "K2", "K4"])
    ap.add_argument("--rays", type=int, default=0, help="rays per GPU per step (0 = the workload's own count)")
    ap.add_argument("--stage", default="color")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the lines under 'extras' (K2, K4 shard, fine stage, operand modes, pipeline flipped; N > 1: K4 at 10000 rays / N, K2, pipeline flipped)")
    ap.add_argument("--no-frustum-mask", action="store_true", help="optimise every voxel (mapping.frustum_feature_selection: False)")
    ap.add_argument("--graph", action="store_true", help="replay each batch's step as a captured hipGraph (single GPU)")
    ap.add_argument("--pipeline", type=int, default=-1, help="1: register the next batch before every step (nsk_map_prepare): its sampling rides in the step's composite "
                    "launch, its cell sort in the backward and Adam launches; 0: every step samples its own batch first; -1 (default): on")
    ap.add_argument("--torch-exchange", action="store_true", help="N > 1: all-reduce the packed buffer through torch.distributed instead of nsk_allreduce_grads")
    ap.add_argument("--tune", action="append", default=[], help="key=value for nsk_set_tuning (experiments), repeatable")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bound of each CPU baseline's timed sample")
    ap.add_argument("--k5-only", action="store_true", help="experiments: only the K5_loop extra (BASELINE configs[4] as a loop), printed as its own JSON line")
    args = ap.parse_args()
    TUNE[:] = args.tune

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU fallback)"
    # rehearsal of the N > 1 path on a one-GPU box: NSK_BENCH_REHEARSE=1 puts every rank on cuda:0 and exchanges over gloo
    # (numbers from such a run mean nothing; the driver's multi-GPU run uses RCCL, one rank per GPU)
    rehearse = os.environ.get("NSK_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    comm = None
    exchange = "none (one GPU)"
    if world > 1:
        import torch.distributed as dist
        import nice_slam_cpp_amd.dist as nd
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        exchange = "torch.distributed all_reduce of the packed buffer (nsk_grad_pack / nsk_grad_unpack around it)"
        if not args.torch_exchange and not rehearse:
            comm = nd.rccl_comm_from_group()                     # ncclUniqueId broadcast over the process group
            if comm is not None:
                exchange = "nsk_allreduce_grads (pack -> ncclAllReduce -> unpack on the context's stream; RCCL communicator bootstrapped from the process group)"
    pipeline = True if args.pipeline < 0 else bool(args.pipeline)
    if args.k5_only:
        k5 = run_k5_loop(local, rank, world, dist, comm, cycles=20, tune=args.tune)
        if rank == 0:
            print(json.dumps(k5))
        if dist is not None:
            dist.destroy_process_group()
        return

    W = workloads()
    wl = W[args.workload]
    N = args.rays or wl["rays"]
    frustum = not args.no_frustum_mask
    res = run_workload(wl, args.stage, N, args.steps, args.warmup, local, rank, world, dist, graph=args.graph, frustum=frustum, pipeline=pipeline, comm=comm,
                       repeats=args.repeats)
    headline_cfg = args.workload == "K3" and N == W["K3"]["rays"] and args.stage == "color" and frustum
    head = summarize(res, args.stage, N, args.steps, world, with_counters=headline_cfg) if rank == 0 else None
    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same (default) command:
    # FETCH_SIZE and WRITE_SIZE collected in separate passes, KB; FETCH_SIZE doubled as the MI355X guide prescribes for gfx950
    pmc_name = _first_profile(PMC_FILES)
    if rank == 0 and pmc_name is not None and headline_cfg:
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
        prefix = KERNEL_PREFIX.get(head["roofline"]["kernel"])
        get = lambda c: next((v for k, v in pmc.items() if prefix and k.startswith(prefix) and k.endswith("|" + c)), None)
        if get("FETCH_SIZE") is not None and get("WRITE_SIZE") is not None:
            head["roofline"]["traffic"] = (2.0 * get("FETCH_SIZE") + get("WRITE_SIZE")) * 1024.0
            head["roofline"]["traffic_source"] = "profiles/" + pmc_name + " (committed FETCH_SIZE / WRITE_SIZE passes of this command, replayed: NOT measured in this run)"
    out = None
    if rank == 0:
        out = {
            "metric": "mapping rays/sec (and ms/iter) on CoFusion room1 at 1/2/4/8 MI355X",
            "value": head["value"], "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "repeats": head["repeats"], "block_ms_per_step": head["block_ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "dtype_detail": "fp32 storage and accumulation; matrix operands as two fp16 pieces (22 significant bits: forward, backward chains) or two "
                                            "bf16 pieces (16 bits: weight-gradient panels); fp32 MFMA for the grid-gradient scatter",
            "data": "synthetic",
            "config": {"workload": "%s; %d rays x 48 samples per GPU, %s-stage mapping iteration (cell sort of the samples + forward + L1 depth/colour loss + "
                                   "backward to the %s grids%s + Adam)" % (wl["name"], N, args.stage, "/".join(STAGE_DECODERS[args.stage]),
                                                                         " and the colour decoder" if args.stage == "color" else ""),
                       "id": args.workload, "rays_per_gpu": N, "samples_per_ray": 48, "stage": args.stage,
                       "grid_shapes": {k: list(v.shape) for k, v in res["sc"]["grids"].items()},
                       "matmul": "fp32 operands as 16-bit pieces on the matrix cores with fp32 accumulation: forward = two fp16 pieces (22 significant bits, "
                                 "3 MFMAs per K=32 block); backward chains (frozen and trainable decoders) = two fp16 pieces of a per-sample power-of-two "
                                 "multiple of the gradient; the trainable decoder's weight-gradient panels = two bf16 pieces (16 bits, fp32 sums); fp32 MFMA "
                                 "for the grid-gradient scatter",
                       "frustum_feature_selection": res["mask_frac"] is not None, "marked_voxel_fraction": res["mask_frac"],
                       "exchange": exchange, "pipeline": pipeline, "launches_per_step": (4 if pipeline else 7) + (3 if world > 1 else 0),
                       "pipeline_detail": ("every timed step holds one batch's sampling + cell sort, forward, loss, backward and Adam; the sampling and sort are those of the "
                                           "NEXT batch, registered before the step (nsk_map_prepare), and ride in this step's composite / backward / Adam launches "
                                           "instead of three launches of their own in front of the forward") if pipeline else "every step samples and sorts its own batch first",
                       "parallelism": "rays sharded x%d, 1 all-reduce/step of %d floats (%.2f MB: marked voxels of the trained levels + colour decoder + loss; "
                                      "the dense gradient slab is %d floats)" % (world, res["exchange_floats"], 4e-6 * res["exchange_floats"], res["slab_floats"])},
            "roofline": head["roofline"], "step_rooflines": head["step_rooflines"], "kernels": head["kernels"], "final_loss": head["final_loss"],
        }
        if "allreduce" in head["kernels"]:
            out["allreduce_us"] = head["kernels"]["allreduce"]["avg_us"]
            out["exchange_bytes"] = 4 * res["exchange_floats"]
    if not args.no_extras:
        # every rank runs the extras (at N > 1 they exchange); rank 0 reports.  N = 1: the other BASELINE configs and the operand modes;
        # N > 1: configs[3] as BASELINE states it (a FIXED 10000-ray batch sharded over the ranks: strong scaling), configs[1], and the
        # headline with the pipeline setting flipped -- the driver's one run per N is the only node time there is
        plan = []
        if world == 1:
            plan += [("K3_fine_stage", "K3", "fine", dict(N=W["K3"]["rays"]), 100), ("K2_color", "K2", "color", dict(N=W["K2"]["rays"]), 300),
                     ("K4_shard_color", "K4", "color", dict(N=W["K4"]["rays"]), 300),
                     ("K3_color_mode0", "K3", "color", dict(N=W["K3"]["rays"], matmul_mode=0), 100),
                     ("K3_color_mode1", "K3", "color", dict(N=W["K3"]["rays"], matmul_mode=1), 100),
                     ("K3_color_backward_fp32", "K3", "color", dict(N=W["K3"]["rays"], backward_mode=0), 100),
                     ("K3_color_no_mask", "K3", "color", dict(N=W["K3"]["rays"], frustum=False), 100),
                     ("K3_color_forward_only", "K3", "color", dict(N=W["K3"]["rays"], forward_only=True, pipeline=False), 200),
                     ("K2_color_forward_only", "K2", "color", dict(N=W["K2"]["rays"], forward_only=True, pipeline=False), 300),
                     ("K3_pipeline_%s" % ("off" if pipeline else "on"), "K3", "color", dict(N=W["K3"]["rays"], pipeline=not pipeline), 200)]
        else:
            plan += [("K4_strong_10000_rays", "K4", "color", dict(N=0, rays_total=10000), 300), ("K2_color", "K2", "color", dict(N=W["K2"]["rays"]), 300),
                     ("K3_pipeline_%s" % ("off" if pipeline else "on"), "K3", "color", dict(N=W["K3"]["rays"], pipeline=not pipeline), args.steps)]
        extras = {}
        for name, wname, stage, kw, k in plan:
            n = kw.pop("N")
            if world == 1 and wname == args.workload and stage == args.stage and n == N and not kw:
                continue
            kw.setdefault("frustum", frustum)
            kw.setdefault("pipeline", pipeline)
            r = run_workload(W[wname], stage, n, k, 20, local, rank, world, dist, comm=comm, **kw)
            if rank != 0:
                continue
            if kw.get("forward_only"):                               # no backward launch to price: the line is time and launches only
                ms = 1e3 * r["dt"] / k
                extras[name] = {"workload": W[wname]["name"], "stage": stage, "rays_per_gpu": r["rays_per_gpu"], "steps": k,
                                "value": r["rays_per_gpu"] * world / (1e-3 * ms), "unit": "rays/s", "ms_per_step": ms,
                                "kernels_avg_us": {kk: round(1e3 * t / c, 2) for kk, (c, t) in r["prof"].items()},
                                "what": "forward only: nsk_render_forward = Renderer::render_batch_ray (z sampling, cell sort where it pays, all decoders of the "
                                        "stage, compositing; no loss, backward or Adam) -- SURVEY 8(d)'s forward-only rays/s"}
                continue
            sm = summarize(r, stage, r["rays_per_gpu"], k, world)
            extras[name] = extra_line(name, W[wname], stage, r, sm, k)
            if kw.get("rays_total"):
                extras[name]["scaling"] = "strong"
                extras[name]["rays_total"] = kw["rays_total"]
                extras[name]["value"] = kw["rays_total"] / (1e-3 * sm["ms_per_step"])
            if "backward_mode" in kw:
                extras[name]["backward_chains"] = "fp32 MFMA (v_mfma_f32_16x16x4_f32) in every role: nsk_set_backward_mode(0), a measuring stick"
            if "matmul_mode" in kw:
                extras[name]["forward_operands"] = {0: "fp32 MFMA (v_mfma_f32_16x16x4_f32)", 1: "three bf16 pieces (24 bits)"}[kw["matmul_mode"]]
        # BASELINE configs[4]: the Tracker + Mapper loop with bundle adjustment (at every N: on one GPU the same code with world = 1)
        k5 = run_k5_loop(local, rank, world, dist, comm, cycles=10 if world == 1 else 20)
        if rank == 0:
            extras["K5_loop"] = k5
            out["extras"] = extras
    if rank != 0:
        if dist is not None:
            if comm is not None:
                nd.rccl_comm_destroy(comm)
            dist.destroy_process_group()
        return
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_aten(res["sc"], res["pool"], args.stage, res["lr"], res["w_color"], args.cpu_seconds)
        out["cpu_baseline_c_port"] = cpu_baseline_c(res["sc"], res["pool"], args.stage, res["lr"], res["w_color"], args.cpu_seconds)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if dist is not None:
        if comm is not None:
            nd.rccl_comm_destroy(comm)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
