#!/usr/bin/env python3
"""Benchmark of the hot path: one mapping iteration (render forward + Mapper loss + backward + [all-reduce] +
Adam) per step, on BASELINE.json configs[1]: config/nice_slam.yaml grids (Replica-room0-class bound, 3-level
grid), 1000 rays x 48 samples per GPU, colour stage.  Prints ONE JSON line (rank 0).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rays shard across ranks with no data-path exchange except ONE all-reduce (RCCL, sum, fp32) of the gradient
slab per step; per-GPU work is fixed, so scaling is "weak".
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
PEAK_HBM_GBS = 8000.0              # same guide, HBM3E peak

# algorithmic MACs per sample of each kernel (DESIGN.md "Kernels")
MAC = {
    "decode_fwd_coarse": 6176, "decode_fwd_middle": 15479, "decode_fwd_fine": 20599, "decode_fwd_color": 15575,
    "decode_bwd_coarse": 6176, "decode_bwd_middle": 9248, "decode_bwd_fine": 9248, "decode_bwd_color": 9344,
    # trainable: input gradients + embedding-gradient products + weight gradients (block outputs come from the forward, which
    # saved them: no recompute; the fine decoder still recomputes its forward)
    "decode_bwd_coarse_train": 6176 * 2, "decode_bwd_middle_train": 9248 + 5952 + 15479,
    "decode_bwd_fine_train": 20599 + 9248 + 5952 + 20599, "decode_bwd_color_train": 9344 + 5952 + 15575,
}
# algorithmic HBM bytes per sample of each kernel: 8 corners x 32 ch x 4 B per level read, 2x that per level
# scattered (read-modify-write), plus per-sample intermediates
BYTES = {
    "decode_fwd_coarse": 1024 + 8, "decode_fwd_middle": 1024 + 8, "decode_fwd_fine": 2048 + 8, "decode_fwd_color": 1024 + 20,
    "decode_bwd_coarse": 2048 + 20, "decode_bwd_middle": 2048 + 52, "decode_bwd_fine": 2048 + 52, "decode_bwd_color": 2048 + 52,
    "decode_bwd_coarse_train": 3072 + 20 + 640, "decode_bwd_middle_train": 3072 + 20 + 640, "decode_bwd_fine_train": 4096 + 20,
    "decode_bwd_color_train": 3072 + 20 + 640,          # + the 160 saved block outputs read back
}


def cpu_baseline(sc, rays_list, lr, w_color, seconds=12.0):
    """the oracle (a port of the reference's path, oracle/nso.c) timed on this host: same step, same inputs"""
    from oracle.nso import Oracle
    o = Oracle("f32")
    threads = int(o.lib.nso_num_threads())
    grids = {k: v.copy() for k, v in sc["grids"].items()}
    decs = {k: v.copy() for k, v in sc["decoders"].items()}
    mom = {k: (np.zeros_like(grids[k]), np.zeros_like(grids[k])) for k in ("middle", "fine", "color")}
    dm, dv = np.zeros_like(decs["color"]), np.zeros_like(decs["color"])
    op = o.opts(sc["bound"])
    n_rays, steps, t0 = 0, 0, time.perf_counter()
    while True:
        r = rays_list[steps % len(rays_list)]
        fw = o.render_forward(op, grids, decs, "color", r["rays_o"], r["rays_d"], r["gt_depth"])
        _, g_d, g_c = o.loss_map(fw["depth"], fw["rgb"], r["gt_depth"], r["gt_color"], w_color, True)
        bw = o.render_backward(op, grids, decs, "color", r["rays_o"], r["rays_d"], r["gt_depth"], -1.0, g_c, g_d, None,
                               want_rays=False)
        steps += 1
        for k in mom:
            o.adam_step(grids[k], bw["g_grids"][k], mom[k][0], mom[k][1], lr[2], steps)
        o.adam_step(decs["color"], bw["g_decoders"]["color"], dm, dv, lr[0], steps)
        n_rays += r["rays_o"].shape[0]
        if time.perf_counter() - t0 > seconds or steps >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": n_rays / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": "%d full mapping steps of %d rays x 48 samples (colour stage, same scene), %.1f s, OpenMP over rays"
                      % (steps, rays_list[0]["rays_o"].shape[0], dt), "ms_per_step": 1e3 * dt / steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=1000, help="rays per GPU per step (config/nice_slam.yaml mapping.pixels)")
    ap.add_argument("--stage", default="color")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay each batch's step as a captured hipGraph (single GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import nice_slam_cpp_amd as pkg
    import scenes

    N = args.rays
    sc = scenes.make_scene(42)                                   # reference grid shapes + init (src/main.cpp:33-78)
    pool = [scenes.make_rays(1234 + 17 * i + 1000 * rank, N, sc["bound"], n_frames=5) for i in range(8)]
    ctx = pkg.Context(local)
    ctx.set_render_opts()                                        # 32 + 16 samples (src/Renderer.cpp:9-10)
    ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"])
    ctx.decoder_set_trainable("color", True)                     # fix_fine: True, fix_color: False (nice_slam.yaml:51-52)
    lr = [0.005, 0.0, 0.005, 0.005, 0.005, 0.0]                  # mapping.stage.color (nice_slam.yaml:90-95), lr_factor 1
    w_color = 0.5                                                # src/Mapper.cpp:33 reads tracking.w_color_loss (D20)
    dev = torch.device("cuda", local)
    cu = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev).contiguous()
    batches = []
    import nice_slam_cpp_amd.dist as nd
    for r in pool:
        gd = cu(r["gt_depth"])
        gmax = nd.global_depth_max(gd)                           # batch-global max(gt_depth) (Renderer.cpp:76,93)
        batches.append((cu(r["rays_o"]), cu(r["rays_d"]), gd, cu(r["gt_color"]), gmax))
    loss = torch.zeros(1, device=dev)
    flags = pkg.nsk.GRAD_GRIDS | pkg.nsk.GRAD_DECODERS

    with torch.cuda.stream(ctx.tstream):
        def step(i):
            ro, rd, gd, gc, gmax = batches[i % len(batches)]
            ctx.map_step(args.stage, ro, rd, gd, gc, gmax, w_color, True, flags=flags, loss=loss)
            if world > 1:                                        # the one exchange of the path; grad_slab() completes the
                nd.allreduce_grads(ctx.grad_slab())              # step's pending gradient reductions before it is read
            ctx.adam_step(lr)

        if args.graph and world == 1:                           # one graph per batch of the pool, replayed instead of re-issued
            step(0)                                              # sizes the workspaces
            gids = []
            for b in range(len(batches)):
                ctx.graph_begin(); step(b); gids.append(ctx.graph_end())
            eager_step = step

            def step(i):                                         # noqa: F811
                ctx.graph_launch(gids[i % len(gids)])
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        # per-kernel durations: HIP events recorded on the context's stream around every launch (same steps again)
        ctx.profile_begin()
        for i in range(args.steps):
            (eager_step if args.graph and world == 1 else step)(args.warmup + i)      # events are recorded around eager launches
        prof = ctx.profile_end()
        final_loss = float(loss)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    S = 48
    M = N * S
    per_kernel = {k: {"launches": c, "avg_us": 1e3 * ms / c} for k, (c, ms) in prof.items()}
    # merged launches: all decoders of the stage run as workgroup roles of one kernel
    MAC["decode_fwd_multi"] = MAC["decode_fwd_middle"] + MAC["decode_fwd_fine"] + (MAC["decode_fwd_color"] if args.stage == "color" else 0)
    BYTES["decode_fwd_multi"] = BYTES["decode_fwd_middle"] + BYTES["decode_fwd_fine"] + (BYTES["decode_fwd_color"] + 640 if args.stage == "color" else 0)
    MAC["decode_bwd_multi"] = MAC["decode_bwd_middle"] + MAC["decode_bwd_fine"] + (MAC["decode_bwd_color_train"] if args.stage == "color" else 0)
    BYTES["decode_bwd_multi"] = BYTES["decode_bwd_middle"] + BYTES["decode_bwd_fine"] + (BYTES["decode_bwd_color_train"] if args.stage == "color" else 0)
    dom = max((k for k in prof if k.startswith("decode")), key=lambda k: prof[k][1])
    dom_s = prof[dom][1] / prof[dom][0] * 1e-3
    flops = 2.0 * MAC[dom] * M
    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    # (profiles/r01f_pmc_hbm.json, tools/profile_round.sh: FETCH_SIZE and WRITE_SIZE collected in separate passes, KB; FETCH_SIZE doubled as the
    # MI355X guide prescribes for gfx950).  Only valid for the default 1000-ray workload the passes were run on.
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01f_pmc_hbm.json")
    if os.path.exists(pmc_path) and N == 1000 and args.stage == "color":
        pmc = json.load(open(pmc_path))
        prefix = {"decode_bwd_multi": "void k_decode_bwd_multi<false>", "decode_fwd_multi": "k_decode_fwd_multi_bf16"}.get(dom)
        get = lambda c: next((v for k, v in pmc.items() if prefix and k.startswith(prefix) and k.endswith("|" + c)), None)
        if get("FETCH_SIZE") is not None and get("WRITE_SIZE") is not None:
            traffic = (2.0 * get("FETCH_SIZE") + get("WRITE_SIZE")) * 1024.0
    roof = {"bound": "mfma", "kernel": dom, "achieved": flops / dom_s / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": flops / dom_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
            "avg_launch_us": dom_s * 1e6, "alg_flops_per_launch": flops,
            "alg_bytes_per_launch": BYTES[dom] * M, "hbm_frac_same_kernel": BYTES[dom] * M / dom_s / 1e9 / PEAK_HBM_GBS}
    step_bytes = (9221.0 + 1280.0) * M + 28.0 * sum(sc["grids"][k].size for k in ("middle", "fine", "color"))   # + saved block outputs, written and read
    step_flops = 2.0 * (MAC["decode_fwd_multi"] + MAC["decode_bwd_multi"]) * M + 1000.0 * M                    # + sampling / compositing
    out = {
        "metric": "mapping rays/sec (and ms/iter) on CoFusion room1 at 1/2/4/8 MI355X",
        "value": world * N * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: config/nice_slam.yaml grids (bound of src/main.cpp:33, 3-level grid + colour), "
                               "%d rays x 48 samples per GPU, colour-stage mapping iteration "
                               "(forward + L1 depth/colour loss + backward to middle/fine/colour grids and colour decoder + Adam)" % N,
                   "rays_per_gpu": N, "samples_per_ray": S, "stage": args.stage,
                   "matmul": "forward: fp32 operands as 3 bf16 pieces, 6 bf16 MFMAs per product (fp32-accurate); backward: fp32 MFMA", "parallelism": "rays sharded x%d, 1 all-reduce/step" % world},
        "roofline": roof,
        "step_rooflines": {"alg_bytes_per_step": step_bytes, "hbm_frac": step_bytes / (dt / args.steps) / 1e9 / PEAK_HBM_GBS,
                           "alg_flops_per_step": step_flops, "fp32_frac": step_flops / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
        "kernels": per_kernel, "final_loss": final_loss,
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(sc, pool, lr, w_color, args.cpu_seconds)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
