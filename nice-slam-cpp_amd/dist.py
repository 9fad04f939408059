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

    def __init__(self, backend, group=None):
        self.backend, self.group = backend, group
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
    if lib is None:
        return None

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    uid = UniqueId()
    ok = torch.ones(1, dtype=torch.int32, device="cuda")
    if rank == 0 and lib.ncclGetUniqueId(C.byref(uid)) != 0:
        ok.zero_()
    t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device="cuda")
    dist.broadcast(t, src=0, group=group)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok) == 0:
        return None
    C.memmove(C.byref(uid), bytes(t.cpu().tolist()), 128)
    comm = C.c_void_p()
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rc = lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
    ok.fill_(1 if rc == 0 and comm.value else 0)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)          # all ranks take the same path
    return comm if int(ok) == 1 else None
