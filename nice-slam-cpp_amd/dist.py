"""Multi-GPU mapping: rays shard across ranks, grids / decoders / Adam state are replicated, and each mapping step
needs exactly ONE all-reduce (sum, fp32) of the gradient slab.  `torch.distributed` is the
transport: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

The only couplings that break naive ray sharding are batch-global statistics inside the renderer:
max(gt_depth) (reference src/Renderer.cpp:76,93) -> all-reduced (max) here and passed to the kernels as
`gt_depth_max`; the Tracker's median (src/Tracker.cpp:70) -> tracking stays on one rank (200 rays)."""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """contiguous shard [lo, hi) of n rays for `rank`; shards differ by at most one ray"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_depth_max(gt_depth, group=None):
    """max(gt_depth) over the WHOLE batch although each rank only holds a shard"""
    m = gt_depth.max().reshape(1).clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    return float(m)


def _world(group=None):
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def allreduce_grads(slab, group=None):
    """the one exchange of the path: sum the gradient slab (grids + decoders + loss scalar) over ranks, in place"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(slab, op=dist.ReduceOp.SUM, group=group)
    return slab


class ShardedMapper:
    """One mapping iteration on a ray shard.  `backend` is an nsk Context (or any object with map_step / adam_step /
    grad_slab of the same meaning); every rank ends each step with bit-identical parameters because every rank applies
    the same Adam update to the same all-reduced gradients."""

    def __init__(self, backend, group=None, comm=None):
        """comm: an RCCL communicator (rccl_comm_from_group) -> the exchange is nsk_allreduce_grads on the context's stream; None -> torch.distributed"""
        self.backend, self.group, self.comm = backend, group, comm
        self._slab = None

    def step(self, stage, rays_o, rays_d, gt_depth, gt_color, lr, w_color=0.2, use_color=True, flags=3, loss=None,
             gt_depth_max=None):
        if gt_depth_max is None:
            gt_depth_max = global_depth_max(gt_depth, self.group)
        self.backend.map_step(stage, rays_o, rays_d, gt_depth, gt_color, gt_depth_max, w_color, use_color, flags=flags, loss=loss)
        if _world(self.group) > 1:
            if hasattr(self.backend, "grad_pack"):
                # compact exchange: only the voxels the (rank-identical) optimiser masks mark, only the levels the stage touched
                allreduce_grads(self.backend.grad_pack(), self.group)
                self.backend.grad_unpack()
            else:
                # grad_slab() also completes the step's pending gradient reductions: call it every step, before the exchange
                allreduce_grads(self.backend.grad_slab(), self.group)
        self.backend.adam_step(lr)
        return gt_depth_max


    def step_ba(self, stage, full, frames, cams, cam_m, cam_v, intr, lr, ba_lr, ba_step, xt, g_ro, g_rd, w_color=0.2, use_color=True, mode=0):
        """One BUNDLE-ADJUSTMENT iteration (reference src/Mapper.cpp:305-329,366-368,430-446,467-489) on this rank's contiguous shard
        [lo, hi) of the window's frame-major batch.

        full    the WHOLE batch on every rank (rays_o, rays_d, gt_depth, gt_color, pix_i, pix_j, keep): the pixel draw is a counter hash of the
                seed, so every rank generates the same batch with one small launch (nsk_prepare_rays) and no collective is needed for it, nor for
                max(gt_depth) (src/Renderer.cpp:76,93), which is taken over `full` (set_depth_max_batch);
        frames  per window frame (first, count, active): its ray range in `full` and whether its pose is optimised;
        cams, cam_m, cam_v   [frames, 8] pose 7-vectors and their Adam moments, replicated;
        xt      [8 frames + 8] scratch that travels with the exchange: pose gradients | loss | kept rays | 0 x 6;
        g_ro, g_rd   [N, 3] scratch for the ray gradients (only [lo, hi) is written).
        ONE all-reduce carries the marked voxels' gradients, the trainable decoder's, and xt; every rank then applies the same Adam steps to
        grids, decoder and poses.  Returns (lo, hi)."""
        world, rank = _world(self.group), (dist.get_rank(self.group) if _world(self.group) > 1 else 0)
        N = full["gt_depth"].shape[0]
        lo, hi = shard_range(N, rank, world)
        nf = len(frames)
        be = self.backend
        be.set_depth_max_batch(full["gt_depth"], full["keep"])
        be.set_ray_mask(full["keep"][lo:hi])
        if hi > lo:
            be.map_step(stage, full["rays_o"][lo:hi], full["rays_d"][lo:hi], full["gt_depth"][lo:hi], full["gt_color"][lo:hi], -1.0, w_color, use_color,
                        flags=7, loss=xt[8 * nf:8 * nf + 1], g_rays=(g_ro[lo:hi], g_rd[lo:hi]))
        be.set_ray_mask(None)
        be.set_depth_max_batch(None, None)
        first = [max(lo, f) for f, c, a in frames]                              # the part of every frame's rays that falls into this shard
        count = [max(0, min(hi, f + c) - max(lo, f)) for f, c, a in frames]
        active = [1 if a else 0 for f, c, a in frames]
        be.pose_step_multi(first, count, active, full["pix_i"], full["pix_j"], intr, g_ro, g_rd, cams, step=0, mode=mode, g_cams=xt, keep=full["keep"][lo:hi])
        be.grad_extra(xt)
        if world > 1:
            if self.comm is not None:
                be.allreduce_grads_rccl(self.comm)
            else:
                allreduce_grads(be.grad_pack(), self.group)
                be.grad_unpack()
        be.grad_extra(None)
        be.adam_step(lr)
        be.adam_vector(cams.reshape(-1), xt[:8 * nf], cam_m.reshape(-1), cam_v.reshape(-1), ba_lr, ba_step)
        return lo, hi


def _rccl_lib():
    import ctypes as C
    for name in ("librccl.so", "librccl.so.1"):
        try:
            return C.CDLL(name)
        except OSError:
            pass
    return None


def rccl_comm_from_group(group=None):
    """An RCCL communicator of the process group's ranks for the C-ABI exchange (nsk_allreduce_grads: pack -> ncclAllReduce -> unpack on
    the context's stream, no Python between them): rank 0 draws the ncclUniqueId, the process group broadcasts its 128 bytes, every
    rank calls ncclCommInitRank on its current device.  Returns the communicator as a ctypes void pointer, or None where that cannot
    work (no process group, a CPU backend, librccl absent or an initialisation error) -- the caller then uses `allreduce_grads`."""
    import ctypes as C
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_backend(group) != "nccl":
        return None
    lib = _rccl_lib()

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    uid = UniqueId()
    # every rank reaches the collectives below whatever happened locally (a rank that returned early on a failed dlopen would leave the others
    # blocked in the broadcast): the outcome of the library load and of ncclGetUniqueId is agreed on first
    ok = torch.ones(1, dtype=torch.int32, device="cuda")
    if lib is None or (rank == 0 and lib.ncclGetUniqueId(C.byref(uid)) != 0):
        ok.zero_()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok) == 0:
        return None
    t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device="cuda")
    dist.broadcast(t, src=0, group=group)
    C.memmove(C.byref(uid), bytes(t.cpu().tolist()), 128)
    comm = C.c_void_p()
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rc = lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
    ok.fill_(1 if rc == 0 and comm.value else 0)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)          # all ranks take the same path
    return comm if int(ok) == 1 else None


def rccl_comm_destroy(comm):
    """ncclCommDestroy for a communicator made by rccl_comm_from_group (bench.py calls it before the process group goes away)"""
    lib = _rccl_lib()
    if lib is not None and comm is not None and comm.value:
        lib.ncclCommDestroy.argtypes = [__import__("ctypes").c_void_p]
        lib.ncclCommDestroy(comm)
