"""ctypes binding of the C-ABI in include/nsk.h (libnsk.so, built from csrc/ for gfx950).

There is no CPU fallback: if the HIP library is missing or no MI355X is present, creating a context raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("NSK_LIB", os.path.join(_HERE, "csrc", "libnsk.so"))

STAGES = {"coarse": 0, "middle": 1, "fine": 2, "color": 3}
LEVELS = ("coarse", "middle", "fine", "color")
GRAD_GRIDS, GRAD_DECODERS, GRAD_RAYS = 1, 2, 4
GROUP_DECODERS, GROUP_COARSE, GROUP_MIDDLE, GROUP_FINE, GROUP_COLOR, GROUP_CAMERA = range(6)

# every symbol include/nsk.h declares
SYMBOLS = (
    "nsk_last_error", "nsk_version", "nsk_ctx_create", "nsk_ctx_destroy", "nsk_sync", "nsk_stream", "nsk_set_bound",
    "nsk_set_render_opts", "nsk_set_matmul_mode", "nsk_set_sort_mode", "nsk_set_tuning", "nsk_set_ray_mask", "nsk_grid_upload", "nsk_grid_download", "nsk_grid_grad_download", "nsk_set_mask", "nsk_frustum_mask", "nsk_keyframe_overlap", "nsk_sample_pixels", "nsk_gather_pixels", "nsk_rays_from_camera", "nsk_pose_step",
    "nsk_decoder_param_count", "nsk_decoder_upload", "nsk_decoder_download", "nsk_decoder_grad_download",
    "nsk_decoder_set_trainable", "nsk_render_forward", "nsk_eval_points", "nsk_raw2outputs", "nsk_render_backward", "nsk_map_step",
    "nsk_track_step", "nsk_loss_map", "nsk_loss_track", "nsk_rays_from_pixels", "nsk_rays_backward",
    "nsk_camera_from_tensor", "nsk_camera_backward", "nsk_inside_filter", "nsk_adam_vector", "nsk_adam_step",
    "nsk_adam_reset", "nsk_graph_begin", "nsk_graph_end", "nsk_graph_launch", "nsk_graph_destroy", "nsk_zero_grads", "nsk_prepare_rays", "nsk_map_prepare", "nsk_grad_slab", "nsk_grad_pack", "nsk_grad_unpack", "nsk_allreduce_grads", "nsk_last_call_stats",
    "nsk_profile_begin", "nsk_profile_end", "nsk_debug_relu_bits", "nsk_debug_preact", "nsk_debug_fetch",
    "nsk_pose_step_multi", "nsk_set_depth_max_batch", "nsk_grad_extra", "nsk_set_backward_mode",
)


class NskError(RuntimeError):
    pass


def build(force=False):
    """compile csrc/ for gfx950 (hipcc cross-compiles without a GPU)"""
    # staleness is make's business: csrc/Makefile lists every source and header of the library
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")] + (["-B"] if force else []) + ["all"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise NskError("libnsk.so is not built (%s); run __graft_entry__.build() -- there is no fallback path" % _LIB_PATH)
        try:
            import torch  # noqa: F401  -- first: torch ships its own HIP runtime; when libnsk.so is loaded before it, two runtimes end up in the process and
        except Exception:             # nsk_ctx_create sees no device although torch.cuda does (build() followed by smoke() in one process showed it)
            pass
        L = C.CDLL(_LIB_PATH)
        L.nsk_last_error.restype = C.c_char_p
        L.nsk_decoder_param_count.restype = C.c_size_t
        L.nsk_decoder_param_count.argtypes = [C.c_int]
        L.nsk_stream.restype = C.c_void_p
        L.nsk_stream.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise NskError(lib().nsk_last_error().decode())


def _ptr(t):
    """device pointer of a contiguous torch CUDA tensor (or None)"""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "expected a contiguous CUDA tensor"
    return C.c_void_p(t.data_ptr())


def _stage(s):
    return STAGES[s] if isinstance(s, str) else int(s)


class _CudaArray:
    """__cuda_array_interface__ view of a raw device pointer so that torch can wrap context-owned memory"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def _ordered(fn):
    """order the context stream against torch's current stream around a launching call"""
    import functools

    @functools.wraps(fn)
    def wrap(self, *a, **k):
        import torch
        cur = torch.cuda.current_stream(self.device)
        same = cur == self.tstream
        if not same:
            self.tstream.wait_stream(cur)
        try:
            return fn(self, *a, **k)
        finally:
            if not same:
                cur.wait_stream(self.tstream)
    return wrap


class Context:
    """one nsk_ctx (one GPU, one HIP stream)"""

    def __init__(self, device=0, stream=None):
        """stream: a torch.cuda.Stream to launch on (default: a new one).  Every launching method orders the
        context's stream after torch's current stream and the current stream after the launch, so results are
        safe to consume with ordinary torch ops; run under `with torch.cuda.stream(ctx.tstream)` to avoid the
        two event waits per call."""
        import torch
        self.h = C.c_void_p()
        self.device = int(device)
        lib()
        if torch.cuda.is_available():
            self.tstream = stream if stream is not None else torch.cuda.Stream(self.device)
            handle = C.c_void_p(self.tstream.cuda_stream)
        else:
            self.tstream, handle = None, None
        _chk(lib().nsk_ctx_create(self.device, handle, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().nsk_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration -------------------------------------------------------------------------------
    def set_bound(self, bound):
        import numpy as np
        b = np.ascontiguousarray(np.asarray(bound, dtype=np.float32).reshape(6))
        _chk(lib().nsk_set_bound(self.h, b.ctypes.data_as(C.c_void_p)))

    def set_render_opts(self, n_samples=32, n_surface=16, lindisp=False, perturb=0.0, occupancy=False, seed=0):
        _chk(lib().nsk_set_render_opts(self.h, n_samples, n_surface, int(lindisp), C.c_float(perturb), int(occupancy),
                                       C.c_uint64(seed)))
        self.n_samples, self.n_surface = n_samples, n_surface

    def set_matmul_mode(self, mode):
        _chk(lib().nsk_set_matmul_mode(self.h, int(mode)))

    def set_ray_mask(self, keep):
        """uint8 cuda tensor [N] (or None): rays with keep == 0 take no part in the loss, its gradients and the batch statistics"""
        self._ray_mask = keep                     # keep the tensor alive while the context points at it
        _chk(lib().nsk_set_ray_mask(self.h, _ptr(keep) if keep is not None else None))

    def set_backward_mode(self, mode):
        """0: every backward chain on the fp32 MFMA (full-width operands; a measuring stick), 2 (default): two fp16 pieces"""
        _chk(lib().nsk_set_backward_mode(self.h, int(mode)))

    def set_sort_mode(self, mode):
        """-1 automatic, 0 ray order, 1 cell-sorted (include/nsk.h)"""
        _chk(lib().nsk_set_sort_mode(self.h, int(mode)))

    def set_tuning(self, key, value):
        _chk(lib().nsk_set_tuning(self.h, key.encode(), int(value)))

    def sync(self):
        _chk(lib().nsk_sync(self.h))

    # -- grids / decoders (host numpy in the reference layouts) ------------------------------------------
    def grid_upload(self, level, arr):
        import numpy as np
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32))
        if a.ndim == 5:
            a = a[0]
        Cc, Z, Y, X = a.shape
        _chk(lib().nsk_grid_upload(self.h, _stage(level), a.ctypes.data_as(C.c_void_p), Cc, Z, Y, X))
        if not hasattr(self, "_gshape"):
            self._gshape = {}
        self._gshape[_stage(level)] = (Cc, Z, Y, X)

    def grid_download(self, level, grad=False):
        import numpy as np
        out = np.zeros(self._gshape[_stage(level)], np.float32)
        f = lib().nsk_grid_grad_download if grad else lib().nsk_grid_download
        _chk(f(self.h, _stage(level), out.ctypes.data_as(C.c_void_p)))
        return out

    def set_mask(self, level, mask):
        import numpy as np
        if mask is None:
            _chk(lib().nsk_set_mask(self.h, _stage(level), None))
            return
        m = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        _chk(lib().nsk_set_mask(self.h, _stage(level), m.ctypes.data_as(C.c_void_p)))

    @_ordered
    def frustum_mask(self, level, depth_img, intr, c2w):
        """Mapper::get_mask_from_c2w on the device; installs the mask and returns it as bool [Z,Y,X]"""
        import numpy as np
        H, W = depth_img.shape
        Cc, Z, Y, X = self._gshape[_stage(level)]
        m = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32).reshape(4, 4))
        out = np.zeros(Z * Y * X, np.uint8)
        fx, fy, cx, cy = intr
        _chk(lib().nsk_frustum_mask(self.h, _stage(level), _ptr(depth_img), H, W, C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                    C.c_float(cy), m.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out.reshape(Z, Y, X).astype(bool)

    @_ordered
    def sample_pixels(self, seed, n, H0, H1, W0, W1):
        """raySampler's pixel draw on the device -> (pix_i cols, pix_j rows) int32 cuda tensors"""
        import torch
        pi = torch.empty(n, dtype=torch.int32, device="cuda:%d" % self.device)
        pj = torch.empty_like(pi)
        _chk(lib().nsk_sample_pixels(self.h, C.c_ulonglong(seed), n, H0, H1, W0, W1, _ptr(pi), _ptr(pj)))
        return pi, pj

    @_ordered
    def prepare_rays(self, frames, rays_per_frame, window, intr, mode=0, want_keep=True):
        """nsk_prepare_rays: pixel draw + ground-truth gather + ray generation + inside filter for a list of frames in one launch.
        frames: list of dicts {depth [H,W] cuda, color [H,W,3] cuda or None, pose cuda (12 floats c2w or 7 floats), seed};
        window = (H0, H1, W0, W1); returns dict of cuda tensors (pix_i, pix_j, gt_depth, gt_color, rays_o, rays_d, keep)"""
        import torch

        class FrameRays(C.Structure):
            _fields_ = [("d_depth", C.c_void_p), ("d_color", C.c_void_p), ("d_pose", C.c_void_p), ("pose_is_cam7", C.c_int), ("seed", C.c_ulonglong)]
        nf = len(frames)
        tab = (FrameRays * nf)()
        H, W = frames[0]["depth"].shape
        for k, f in enumerate(frames):
            tab[k].d_depth = f["depth"].data_ptr()
            tab[k].d_color = f["color"].data_ptr() if f.get("color") is not None else None
            tab[k].d_pose = f["pose"].data_ptr()
            tab[k].pose_is_cam7 = 1 if f["pose"].numel() == 7 else 0
            tab[k].seed = int(f["seed"])
        n = nf * rays_per_frame
        dev = frames[0]["depth"].device
        out = dict(pix_i=torch.empty(n, dtype=torch.int32, device=dev), pix_j=torch.empty(n, dtype=torch.int32, device=dev),
                   gt_depth=torch.empty(n, dtype=torch.float32, device=dev), gt_color=torch.empty((n, 3), dtype=torch.float32, device=dev),
                   rays_o=torch.empty((n, 3), dtype=torch.float32, device=dev), rays_d=torch.empty((n, 3), dtype=torch.float32, device=dev),
                   keep=torch.empty(n, dtype=torch.uint8, device=dev) if want_keep else None)
        has_color = all(f.get("color") is not None for f in frames)
        H0, H1, W0, W1 = window
        fx, fy, cx, cy = intr
        _chk(lib().nsk_prepare_rays(self.h, nf, tab, rays_per_frame, H0, H1, W0, W1, H, W, C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), mode,
                                    _ptr(out["pix_i"]), _ptr(out["pix_j"]), _ptr(out["gt_depth"]), _ptr(out["gt_color"]) if has_color else None,
                                    _ptr(out["rays_o"]), _ptr(out["rays_d"]), _ptr(out["keep"]) if want_keep else None))
        return out

    @_ordered
    def gather_pixels(self, pix_i, pix_j, depth_img, color_img=None):
        import torch
        n = pix_i.shape[0]
        H, W = depth_img.shape
        gd = torch.empty(n, dtype=torch.float32, device=depth_img.device)
        gc = torch.empty((n, 3), dtype=torch.float32, device=depth_img.device) if color_img is not None else None
        _chk(lib().nsk_gather_pixels(self.h, n, _ptr(pix_i), _ptr(pix_j), H, W, _ptr(depth_img), _ptr(color_img) if color_img is not None else None,
                                     _ptr(gd), _ptr(gc) if gc is not None else None))
        return gd, gc

    @_ordered
    def keyframe_overlap(self, rays_o, rays_d, gt_depth, intr, HW, c2w_list, n_samples=16):
        """Mapper::keyframe_selection_overlap: fraction of the frame's sample points seen by each keyframe -> float32 [K]"""
        import numpy as np
        m = np.ascontiguousarray(np.asarray(c2w_list, dtype=np.float32).reshape(-1, 16))
        out = np.zeros(m.shape[0], np.float32)
        fx, fy, cx, cy = intr
        _chk(lib().nsk_keyframe_overlap(self.h, rays_o.shape[0], _ptr(rays_o), _ptr(rays_d), _ptr(gt_depth), n_samples, int(HW[0]), int(HW[1]),
                                        C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), m.shape[0], m.ctypes.data_as(C.c_void_p),
                                        out.ctypes.data_as(C.c_void_p)))
        return out

    def decoder_upload(self, which, packed):
        import numpy as np
        p = np.ascontiguousarray(np.asarray(packed, dtype=np.float32).reshape(-1))
        _chk(lib().nsk_decoder_upload(self.h, _stage(which), p.ctypes.data_as(C.c_void_p), C.c_size_t(p.size)))

    def decoder_download(self, which, grad=False):
        import numpy as np
        n = lib().nsk_decoder_param_count(_stage(which))
        out = np.zeros(n, np.float32)
        f = lib().nsk_decoder_grad_download if grad else lib().nsk_decoder_download
        _chk(f(self.h, _stage(which), out.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
        return out

    def decoder_set_trainable(self, which, flag):
        _chk(lib().nsk_decoder_set_trainable(self.h, _stage(which), int(flag)))

    def load_scene(self, bound, grids, decoders):
        self.set_bound(bound)
        for k in LEVELS:
            if k in grids and grids[k] is not None:
                self.grid_upload(k, grids[k])
            if k in decoders and decoders[k] is not None:
                self.decoder_upload(k, decoders[k])

    # -- rendering (torch CUDA tensors in, torch CUDA tensors out) ------------------------------------------
    def _S(self, gt_depth):
        return getattr(self, "n_samples", 32) + (getattr(self, "n_surface", 16) if gt_depth is not None else 0)

    @_ordered
    def render_forward(self, stage, rays_o, rays_d, gt_depth=None, gt_depth_max=-1.0, want_weights=True):
        import torch
        N = rays_o.shape[0]
        dev = rays_o.device
        rgb = torch.empty(N, 3, device=dev); depth = torch.empty(N, device=dev); var = torch.empty(N, device=dev)
        w = torch.empty(N, self._S(gt_depth), device=dev) if want_weights else None
        _chk(lib().nsk_render_forward(self.h, _stage(stage), N, _ptr(rays_o), _ptr(rays_d), _ptr(gt_depth),
                                      C.c_float(gt_depth_max), _ptr(rgb), _ptr(depth), _ptr(var), _ptr(w)))
        return rgb, depth, var, w

    @_ordered
    def eval_points(self, stage, pts):
        import torch
        M = pts.shape[0]
        raw = torch.empty(M, 4, device=pts.device)
        _chk(lib().nsk_eval_points(self.h, _stage(stage), M, _ptr(pts), _ptr(raw)))
        return raw

    @_ordered
    def raw2outputs(self, raw, z, rays_d, occupancy=False):
        import torch
        N, S = z.shape
        dev = z.device
        rgb = torch.empty(N, 3, device=dev); depth = torch.empty(N, device=dev); var = torch.empty(N, device=dev)
        w = torch.empty(N, S, device=dev)
        _chk(lib().nsk_raw2outputs(self.h, N, S, _ptr(raw), _ptr(z), _ptr(rays_d), int(occupancy), _ptr(rgb), _ptr(depth),
                                   _ptr(var), _ptr(w)))
        return rgb, depth, var, w

    @_ordered
    def render_backward(self, stage, rays_o, rays_d, gt_depth, gt_depth_max, g_rgb, g_depth, g_var=None, flags=GRAD_GRIDS):
        import torch
        N = rays_o.shape[0]
        g_ro = g_rd = None
        if flags & GRAD_RAYS:
            g_ro = torch.empty(N, 3, device=rays_o.device); g_rd = torch.empty(N, 3, device=rays_o.device)
        _chk(lib().nsk_render_backward(self.h, _stage(stage), N, _ptr(rays_o), _ptr(rays_d), _ptr(gt_depth),
                                       C.c_float(gt_depth_max), _ptr(g_rgb), _ptr(g_depth), _ptr(g_var), C.c_uint(flags),
                                       _ptr(g_ro), _ptr(g_rd)))
        return g_ro, g_rd

    @_ordered
    def map_step(self, stage, rays_o, rays_d, gt_depth, gt_color, gt_depth_max=-1.0, w_color=0.2, use_color=True,
                 flags=GRAD_GRIDS | GRAD_DECODERS, loss=None, outputs=None, g_rays=None):
        N = rays_o.shape[0]
        rgb, depth, var = outputs if outputs is not None else (None, None, None)
        g_ro, g_rd = g_rays if g_rays is not None else (None, None)
        _chk(lib().nsk_map_step(self.h, _stage(stage), N, _ptr(rays_o), _ptr(rays_d), _ptr(gt_depth), _ptr(gt_color),
                                C.c_float(gt_depth_max), C.c_float(w_color), int(use_color), C.c_uint(flags), _ptr(loss),
                                _ptr(rgb), _ptr(depth), _ptr(var), _ptr(g_ro), _ptr(g_rd)))

    @_ordered
    def map_prepare(self, stage, rays_o, rays_d, gt_depth, gt_depth_max=-1.0, flags=GRAD_GRIDS | GRAD_DECODERS):
        """nsk_map_prepare: register the NEXT batch before the current batch's map_step; its sampling and cell sort then ride in that step's
        composite / backward / Adam launches (include/nsk.h).  The ray mask remembered for the batch is the BUFFER installed by set_ray_mask
        when this is called; it is read later (inside the current step, or at the batch's own step), so it must not be the buffer the current
        step's mask lives in and must stay unchanged until the batch's step has run."""
        self._prep_keep = (getattr(self, "_ray_mask", None), rays_o, rays_d, gt_depth)      # alive until the next registration
        _chk(lib().nsk_map_prepare(self.h, _stage(stage), rays_o.shape[0], _ptr(rays_o), _ptr(rays_d), _ptr(gt_depth), C.c_float(gt_depth_max), C.c_uint(flags)))

    @_ordered
    def track_step(self, stage, rays_o, rays_d, gt_depth, gt_color, gt_depth_max=-1.0, w_color=0.5, use_color=True,
                   handle_dynamic=True, detach_var=True, flags=GRAD_RAYS, loss=None, g_rays=None):
        N = rays_o.shape[0]
        g_ro, g_rd = g_rays if g_rays is not None else (None, None)
        _chk(lib().nsk_track_step(self.h, _stage(stage), N, _ptr(rays_o), _ptr(rays_d), _ptr(gt_depth), _ptr(gt_color),
                                  C.c_float(gt_depth_max), C.c_float(w_color), int(use_color), int(handle_dynamic),
                                  int(detach_var), C.c_uint(flags), _ptr(loss), _ptr(g_ro), _ptr(g_rd)))

    @_ordered
    def loss_map(self, depth, rgb, gt_depth, gt_color, w_color, use_color):
        import torch
        N = depth.shape[0]
        g_d = torch.empty(N, device=depth.device); g_c = torch.empty(N, 3, device=depth.device)
        loss = torch.zeros(1, device=depth.device)
        _chk(lib().nsk_loss_map(self.h, N, _ptr(depth), _ptr(rgb), _ptr(gt_depth), _ptr(gt_color), C.c_float(w_color),
                                int(use_color), _ptr(g_d), _ptr(g_c), _ptr(loss)))
        return loss, g_d, g_c

    @_ordered
    def loss_track(self, depth, rgb, var, gt_depth, gt_color, w_color, use_color, handle_dynamic, detach_var=True):
        import torch
        N = depth.shape[0]
        dev = depth.device
        g_d = torch.empty(N, device=dev); g_c = torch.empty(N, 3, device=dev); g_v = torch.empty(N, device=dev)
        loss = torch.zeros(1, device=dev)
        _chk(lib().nsk_loss_track(self.h, N, _ptr(depth), _ptr(rgb), _ptr(var), _ptr(gt_depth), _ptr(gt_color),
                                  C.c_float(w_color), int(use_color), int(handle_dynamic), int(detach_var), _ptr(g_d),
                                  _ptr(g_c), _ptr(g_v), _ptr(loss)))
        return loss, g_d, g_c, g_v

    # -- rays / pose ------------------------------------------------------------------------------------------
    @_ordered
    def rays_from_pixels(self, pix_i, pix_j, intr, c2w, mode=0):
        import torch
        n = pix_i.shape[0]
        ro = torch.empty(n, 3, device=c2w.device); rd = torch.empty(n, 3, device=c2w.device)
        fx, fy, cx, cy = intr
        _chk(lib().nsk_rays_from_pixels(self.h, n, _ptr(pix_i), _ptr(pix_j), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                        C.c_float(cy), _ptr(c2w), mode, _ptr(ro), _ptr(rd)))
        return ro, rd

    @_ordered
    def rays_from_camera(self, pix_i, pix_j, intr, cam, mode=0):
        """camera_from_tensor + rays_from_pixels in one launch -> (rays_o, rays_d)"""
        import torch
        n = pix_i.shape[0]
        ro = torch.empty(n, 3, device=cam.device); rd = torch.empty(n, 3, device=cam.device)
        fx, fy, cx, cy = intr
        _chk(lib().nsk_rays_from_camera(self.h, n, _ptr(pix_i), _ptr(pix_j), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                        _ptr(cam), mode, _ptr(ro), _ptr(rd), None))
        return ro, rd

    @_ordered
    def pose_step(self, pix_i, pix_j, intr, g_ro, g_rd, cam, m, v, lr, step, mode=0, b1=0.9, b2=0.999, eps=1e-8, g_cam_out=None):
        """rays_backward + camera_backward + adam_vector on the pose in one launch (cam, m, v updated in place)"""
        fx, fy, cx, cy = intr
        _chk(lib().nsk_pose_step(self.h, pix_i.shape[0], _ptr(pix_i), _ptr(pix_j), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                 mode, _ptr(g_ro), _ptr(g_rd), _ptr(cam), _ptr(m), _ptr(v), C.c_float(lr), C.c_float(b1), C.c_float(b2),
                                 C.c_float(eps), int(step), _ptr(g_cam_out)))

    @_ordered
    def pose_step_multi(self, first, count, active, pix_i, pix_j, intr, g_ro, g_rd, cams, m=None, v=None, lr=0.0, step=0, mode=0, b1=0.9, b2=0.999,
                        eps=1e-8, g_cams=None, keep=None):
        """nsk_pose_step_multi: the pose kernels of every frame of a window in one launch.  first / count / active: per-frame ray ranges of the
        batch arrays and "this pose is optimised" flags (host lists); cams [nf, 8] cuda; step >= 1: gradient + Adam (m, v [nf, 8]);
        step == 0: gradients only into g_cams [8 nf + 8] (the N > 1 form; g_cams[8 nf + 1] = kept rays of `keep`)"""
        import numpy as np
        nf = len(first)
        f = np.ascontiguousarray(first, np.int32); c = np.ascontiguousarray(count, np.int32); a = np.ascontiguousarray(active, np.uint8)
        fx, fy, cx, cy = intr
        _chk(lib().nsk_pose_step_multi(self.h, nf, f.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p),
                                       _ptr(pix_i), _ptr(pix_j), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), mode, _ptr(g_ro), _ptr(g_rd),
                                       _ptr(cams), _ptr(m), _ptr(v), C.c_float(lr), C.c_float(b1), C.c_float(b2), C.c_float(eps), int(step), _ptr(g_cams),
                                       _ptr(keep), int(keep.shape[0]) if keep is not None else 0))

    def set_depth_max_batch(self, gt_depth=None, keep=None):
        """nsk_set_depth_max_batch: steps given gt_depth_max < 0 take max(gt_depth) over THIS batch (the whole batch a shard belongs to)"""
        self._dmax = (gt_depth, keep)
        _chk(lib().nsk_set_depth_max_batch(self.h, _ptr(gt_depth), _ptr(keep), int(gt_depth.shape[0]) if gt_depth is not None else 0))

    def grad_extra(self, buf=None):
        """nsk_grad_extra: a float32 cuda vector (multiple of 4 floats) that travels with grad_pack / grad_unpack / allreduce_grads_rccl"""
        self._xextra = buf
        _chk(lib().nsk_grad_extra(self.h, _ptr(buf), C.c_size_t(buf.numel() if buf is not None else 0)))

    @_ordered
    def rays_backward(self, pix_i, pix_j, intr, g_ro, g_rd, mode=0):
        import torch
        g = torch.empty(3, 4, device=g_ro.device)
        fx, fy, cx, cy = intr
        _chk(lib().nsk_rays_backward(self.h, pix_i.shape[0], _ptr(pix_i), _ptr(pix_j), C.c_float(fx), C.c_float(fy),
                                     C.c_float(cx), C.c_float(cy), mode, _ptr(g_ro), _ptr(g_rd), _ptr(g)))
        return g

    @_ordered
    def camera_from_tensor(self, cam):
        import torch
        c2w = torch.empty(3, 4, device=cam.device)
        _chk(lib().nsk_camera_from_tensor(self.h, _ptr(cam), _ptr(c2w)))
        return c2w

    @_ordered
    def camera_backward(self, cam, g_c2w):
        import torch
        g = torch.empty(7, device=cam.device)
        _chk(lib().nsk_camera_backward(self.h, _ptr(cam), _ptr(g_c2w), _ptr(g)))
        return g

    def inside_filter(self, rays_o, rays_d, gt_depth):
        return self._inside_filter_u8(rays_o, rays_d, gt_depth).bool()       # (the conversion is a torch op: after the streams were ordered)

    @_ordered
    def _inside_filter_u8(self, rays_o, rays_d, gt_depth):
        import torch
        keep = torch.empty(rays_o.shape[0], dtype=torch.uint8, device=rays_o.device)
        _chk(lib().nsk_inside_filter(self.h, rays_o.shape[0], _ptr(rays_o), _ptr(rays_d), _ptr(gt_depth), _ptr(keep)))
        return keep

    @_ordered
    def adam_vector(self, p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-8):
        _chk(lib().nsk_adam_vector(self.h, p.numel(), _ptr(p), _ptr(g), _ptr(m), _ptr(v), C.c_float(lr), C.c_float(b1),
                                   C.c_float(b2), C.c_float(eps), int(step)))

    # -- optimiser / multi-GPU ----------------------------------------------------------------------------------
    @_ordered
    def adam_step(self, lr, b1=0.9, b2=0.999, eps=1e-8):
        arr = (C.c_float * 6)(*[float(x) for x in lr])
        _chk(lib().nsk_adam_step(self.h, arr, C.c_float(b1), C.c_float(b2), C.c_float(eps)))

    @_ordered
    def graph_begin(self):
        """record (instead of run) the kernels of the following calls; call from inside `with torch.cuda.stream(ctx.tstream)`"""
        _chk(lib().nsk_graph_begin(self.h))

    def graph_end(self):
        gid = C.c_int(-1)
        _chk(lib().nsk_graph_end(self.h, C.byref(gid)))
        return gid.value

    def graph_launch(self, gid):
        _chk(lib().nsk_graph_launch(self.h, gid))

    def graph_destroy(self, gid):
        _chk(lib().nsk_graph_destroy(self.h, gid))

    def adam_reset(self):
        _chk(lib().nsk_adam_reset(self.h))

    @_ordered
    def zero_grads(self):
        _chk(lib().nsk_zero_grads(self.h))

    @_ordered
    def grad_slab(self):
        """the contiguous gradient slab as a torch tensor aliasing context memory (for torch.distributed all-reduce).
        nsk_grad_slab launches the pending decoder-gradient reduction on the context's stream, so the call is stream-ordered
        like every other launching method: the caller's stream waits for it before the exchange reads the slab.  The wrapping
        tensor is cached per (pointer, size): no per-step host work beyond the C call."""
        import torch
        p, n = C.c_void_p(), C.c_size_t()
        _chk(lib().nsk_grad_slab(self.h, C.byref(p), C.byref(n)))
        key = (p.value, n.value)
        if getattr(self, "_slab_key", None) != key:
            self._slab_t = torch.as_tensor(_CudaArray(p.value, n.value), device="cuda:%d" % self.device)
            self._slab_key = key
        return self._slab_t

    @_ordered
    def grad_pack(self):
        """the step's exchange buffer (marked voxels of the touched levels + trainable decoders + loss) as a torch tensor aliasing
        context memory: all-reduce it, then call grad_unpack()"""
        import torch
        p, n = C.c_void_p(), C.c_size_t()
        _chk(lib().nsk_grad_pack(self.h, C.byref(p), C.byref(n)))
        key = (p.value, n.value)
        if getattr(self, "_pack_key", None) != key:
            self._pack_t = torch.as_tensor(_CudaArray(p.value, n.value), device="cuda:%d" % self.device)
            self._pack_key = key
        return self._pack_t

    @_ordered
    def grad_unpack(self):
        _chk(lib().nsk_grad_unpack(self.h))

    def allreduce_grads_rccl(self, comm):
        """nsk_allreduce_grads with a raw ncclComm_t (ctypes pointer)"""
        _chk(lib().nsk_allreduce_grads(self.h, comm))

    def profile_begin(self):
        _chk(lib().nsk_profile_begin(self.h))

    def profile_end(self):
        """{kernel name: (launches, total ms)} measured with HIP events on the context's stream"""
        buf = C.create_string_buffer(8192)
        _chk(lib().nsk_profile_end(self.h, buf, C.c_size_t(8192)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            out[name] = (int(cnt), float(ms))
        return out

    # -- test aids -------------------------------------------------------------------------------------
    def debug_relu_bits(self, which, M):
        """[M, 5, 32] bool: the ReLU "input > 0" bits the last step's forward saved for decoder `which`, by sample"""
        import numpy as np
        out = np.zeros((M, 5, 32), np.uint8)
        _chk(lib().nsk_debug_relu_bits(self.h, STAGES[which] if isinstance(which, str) else int(which), int(M), out.ctypes.data_as(C.c_void_p)))
        return out.astype(bool)

    def debug_fetch(self, what, M):
        """per-sample array of the last step's workspace: "occ0".."occ2" [M], "rgb4" [M, 4], "g_raw" [M, 4], "z" [M]"""
        import numpy as np
        code = {"occ0": 0, "occ1": 1, "occ2": 2, "rgb4": 3, "g_raw": 4, "z": 5}[what]
        out = np.zeros((M, 4) if code in (3, 4) else (M,), np.float32)
        _chk(lib().nsk_debug_fetch(self.h, code, int(M), out.ctypes.data_as(C.c_void_p)))
        return out

    @_ordered
    def debug_preact(self, which, rays_o, rays_d, M):
        """[M, 5, 32] float32: the ReLU inputs of decoder `which` over the samples of the last step (current matmul mode)"""
        import torch
        out = torch.zeros((M, 5, 32), dtype=torch.float32, device=rays_o.device)
        _chk(lib().nsk_debug_preact(self.h, STAGES[which] if isinstance(which, str) else int(which), int(rays_o.shape[0]), _ptr(rays_o), _ptr(rays_d), _ptr(out)))
        self.sync()
        return out.cpu().numpy()

    def last_call_stats(self):
        b, f, s = C.c_double(), C.c_double(), C.c_int()
        _chk(lib().nsk_last_call_stats(self.h, C.byref(b), C.byref(f), C.byref(s)))
        return b.value, f.value, s.value
