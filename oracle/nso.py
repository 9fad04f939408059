"""ctypes/numpy binding of the CPU oracle (oracle/nso.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Never imported by the product package.  Parity status: see the header of nso.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
STAGES = {"coarse": 0, "middle": 1, "fine": 2, "color": 3}
LEVELS = ("coarse", "middle", "fine", "color")


def build(force=False):
    libs = [os.path.join(_HERE, "libnso_f32.so"), os.path.join(_HERE, "libnso_f64.so")]
    src = os.path.join(_HERE, "nso.c")
    if force or any(not os.path.exists(l) or os.path.getmtime(l) < os.path.getmtime(src) for l in libs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return libs


class _Opts(C.Structure):
    pass


class _Grid(C.Structure):
    pass


def _stage_id(stage):
    return STAGES[stage] if isinstance(stage, str) else int(stage)


class Oracle:
    """One precision variant of the oracle: Oracle('f32') mirrors the reference's fp32 math,
    Oracle('f64') is the same restatement in double."""

    def __init__(self, prec="f32"):
        build()
        self.dt = np.float32 if prec == "f32" else np.float64
        self.creal = C.c_float if prec == "f32" else C.c_double
        self.lib = C.CDLL(os.path.join(_HERE, "libnso_%s.so" % prec))
        assert self.lib.nso_real_size() == np.dtype(self.dt).itemsize
        R = self.creal

        class Opts(C.Structure):
            _fields_ = [("bound", R * 6), ("n_samples", C.c_int), ("n_surface", C.c_int), ("lindisp", C.c_int),
                        ("perturb", R), ("occupancy", C.c_int), ("seed", C.c_uint64)]

        class Grid(C.Structure):
            _fields_ = [("C", C.c_int), ("Z", C.c_int), ("Y", C.c_int), ("X", C.c_int), ("v", C.c_void_p)]

        self.Opts, self.Grid = Opts, Grid
        self.lib.nso_decoder_param_count.restype = C.c_long
        self.lib.nso_depth_max.restype = R
        self.lib.nso_loss_map.restype = R
        self.lib.nso_loss_track.restype = R
        self.lib.nso_inside_filter.restype = C.c_int

    # -- helpers ---------------------------------------------------------------------------
    def arr(self, x, shape=None):
        a = np.ascontiguousarray(np.asarray(x, dtype=self.dt))
        if shape is not None:
            a = a.reshape(shape)
        return a

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def opts(self, bound, n_samples=32, n_surface=16, lindisp=False, perturb=0.0, occupancy=False, seed=0):
        o = self.Opts()
        b = np.asarray(bound, dtype=self.dt).reshape(6)
        for i in range(6):
            o.bound[i] = b[i]
        o.n_samples, o.n_surface, o.lindisp = n_samples, n_surface, int(lindisp)
        o.perturb, o.occupancy, o.seed = perturb, int(occupancy), seed
        return o

    def param_count(self, which):
        return int(self.lib.nso_decoder_param_count(_stage_id(which)))

    def _grids(self, grids):
        """grids: dict level-name -> ndarray [C,Z,Y,X] (or [1,C,Z,Y,X]). Returns (ctypes array, keepalive)."""
        arr = (self.Grid * 4)()
        keep = []
        for i, name in enumerate(LEVELS):
            if name in grids and grids[name] is not None:
                g = self.arr(grids[name])
                if g.ndim == 5:
                    g = g[0]
                keep.append(g)
                arr[i].C, arr[i].Z, arr[i].Y, arr[i].X = g.shape
                arr[i].v = g.ctypes.data
            else:
                arr[i].C = arr[i].Z = arr[i].Y = arr[i].X = 0
                arr[i].v = None
        return arr, keep

    def _decs(self, decoders):
        ptrs = (C.c_void_p * 4)()
        keep = []
        for i, name in enumerate(LEVELS):
            if name in decoders and decoders[name] is not None:
                d = self.arr(decoders[name]).reshape(-1)
                assert d.size == self.param_count(i), (name, d.size, self.param_count(i))
                keep.append(d)
                ptrs[i] = d.ctypes.data
            else:
                ptrs[i] = None
        return ptrs, keep

    # -- render ----------------------------------------------------------------------------
    def render_forward(self, opts, grids, decoders, stage, rays_o, rays_d, gt_depth=None, gt_depth_max=-1.0,
                       want_aux=False):
        ro, rd = self.arr(rays_o, (-1, 3)), self.arr(rays_d, (-1, 3))
        N = ro.shape[0]
        gd = None if gt_depth is None else self.arr(gt_depth, (N,))
        S = opts.n_samples + (opts.n_surface if gd is not None else 0)
        rgb, depth, var = np.zeros((N, 3), self.dt), np.zeros(N, self.dt), np.zeros(N, self.dt)
        w = np.zeros((N, S), self.dt)
        z = np.zeros((N, S), self.dt) if want_aux else None
        raw = np.zeros((N, S, 4), self.dt) if want_aux else None
        ga, k1 = self._grids(grids)
        da, k2 = self._decs(decoders)
        rc = self.lib.nso_render_forward(C.byref(opts), ga, da, _stage_id(stage), N, self._p(ro), self._p(rd),
                                         self._p(gd), self.creal(gt_depth_max), self._p(rgb), self._p(depth),
                                         self._p(var), self._p(w), self._p(z), self._p(raw))
        assert rc == 0
        out = dict(rgb=rgb, depth=depth, var=var, weights=w)
        if want_aux:
            out.update(z=z, raw=raw)
        return out

    def render_backward(self, opts, grids, decoders, stage, rays_o, rays_d, gt_depth, gt_depth_max, g_rgb, g_depth,
                        g_var=None, want_grids=True, want_decoders=True, want_rays=True, relu=None, sigma_on=None):
        """relu (test aid): dict decoder name -> [N*S, 5, 32] bool, the ReLU branches the backward takes instead of its own;
        sigma_on: [N*S] bool, likewise the branch of relu(sigma) in the compositing"""
        ro, rd = self.arr(rays_o, (-1, 3)), self.arr(rays_d, (-1, 3))
        N = ro.shape[0]
        gd = None if gt_depth is None else self.arr(gt_depth, (N,))
        grgb, gdep = self.arr(g_rgb, (N, 3)), self.arr(g_depth, (N,))
        gvar = None if g_var is None else self.arr(g_var, (N,))
        ga, k1 = self._grids(grids)
        da, k2 = self._decs(decoders)
        gg_ptr, gp_ptr = (C.c_void_p * 4)(), (C.c_void_p * 4)()
        gg, gp = {}, {}
        for i, name in enumerate(LEVELS):
            if want_grids and ga[i].v:
                gg[name] = np.zeros((ga[i].C, ga[i].Z, ga[i].Y, ga[i].X), self.dt)
                gg_ptr[i] = gg[name].ctypes.data
            if want_decoders and da[i]:
                gp[name] = np.zeros(self.param_count(i), self.dt)
                gp_ptr[i] = gp[name].ctypes.data
        gro = np.zeros((N, 3), self.dt) if want_rays else None
        grd = np.zeros((N, 3), self.dt) if want_rays else None
        args = (C.byref(opts), ga, da, _stage_id(stage), N, self._p(ro), self._p(rd),
                self._p(gd), self.creal(gt_depth_max), self._p(grgb), self._p(gdep),
                self._p(gvar), gg_ptr if want_grids else None,
                gp_ptr if want_decoders else None, self._p(gro), self._p(grd))
        if relu is None and sigma_on is None:
            rc = self.lib.nso_render_backward(*args)
        else:
            relu = relu or {}
            S = opts.n_samples + (opts.n_surface if gd is not None else 0)
            rp, keep = (C.c_void_p * 4)(), []
            for i, name in enumerate(LEVELS):
                if name in relu and relu[name] is not None:
                    b = np.ascontiguousarray(np.asarray(relu[name]).astype(np.uint8).reshape(N * S, 5, 32))
                    keep.append(b)
                    rp[i] = b.ctypes.data
            so = None if sigma_on is None else np.ascontiguousarray(np.asarray(sigma_on).astype(np.uint8).reshape(N * S))
            rc = self.lib.nso_render_backward_forced(*args, rp, self._p(so))
        assert rc == 0
        return dict(g_grids=gg, g_decoders=gp, g_rays_o=gro, g_rays_d=grd)


    def preacts(self, opts, grids, decoders, stage, which, rays_o, rays_d, gt_depth=None, gt_depth_max=-1.0):
        """[N*S, 5, 32]: the hidden ReLU inputs of decoder `which` at every sample (test aid, see nso.c)"""
        ro, rd = self.arr(rays_o, (-1, 3)), self.arr(rays_d, (-1, 3))
        N = ro.shape[0]
        gd = None if gt_depth is None else self.arr(gt_depth, (N,))
        S = opts.n_samples + (opts.n_surface if gd is not None else 0)
        ga, k1 = self._grids(grids)
        da, k2 = self._decs(decoders)
        a = np.zeros((N * S, 5, 32), self.dt)
        rc = self.lib.nso_preacts(C.byref(opts), ga, da, _stage_id(stage), _stage_id(which), N, self._p(ro), self._p(rd),
                                  self._p(gd), self.creal(gt_depth_max), self._p(a))
        assert rc == 0
        return a

    def preact_bounds(self, opts, grids, decoders, stage, which, rays_o, rays_d, gt_depth=None, gt_depth_max=-1.0, sin_err=3.2e-7, want=None, geometry_err=True,
                      want_raw0=False, quadrature=False):
        """[N*S, 5, 32]: first-order bound on |ReLU input of an fp32 evaluation - exact| at the samples with want[N*S] set (None: all; zeros
        elsewhere) (test aid, see nso.c nso_preact_bounds).  geometry_err=False: the bound between two fp32 evaluations that share z, p and
        p.B bit for bit.  quadrature: the standard deviation of the probabilistic rounding model instead of the worst case.  want_raw0: also return the bound [N*S] on the decoder's first output (the occupancy)"""
        ro, rd = self.arr(rays_o, (-1, 3)), self.arr(rays_d, (-1, 3))
        N = ro.shape[0]
        gd = None if gt_depth is None else self.arr(gt_depth, (N,))
        S = opts.n_samples + (opts.n_surface if gd is not None else 0)
        ga, k1 = self._grids(grids)
        da, k2 = self._decs(decoders)
        tau = np.zeros((N * S, 5, 32), self.dt)
        wt = None if want is None else np.ascontiguousarray(np.asarray(want).astype(np.uint8).reshape(N * S))
        t0 = np.zeros(N * S, self.dt) if want_raw0 else None
        rc = self.lib.nso_preact_bounds(C.byref(opts), ga, da, _stage_id(stage), _stage_id(which), N, self._p(ro), self._p(rd),
                                        self._p(gd), self.creal(gt_depth_max), self.creal(sin_err), int(geometry_err), int(quadrature), self._p(wt), self._p(tau),
                                        self._p(t0))
        assert rc == 0
        return (tau, t0) if want_raw0 else tau

    def ray_fragility(self, opts, grids, decoders, stage, rays_o, rays_d, gt_depth=None, gt_depth_max=-1.0):
        """min |ReLU input| per ray (test aid, see nso.c)"""
        ro, rd = self.arr(rays_o, (-1, 3)), self.arr(rays_d, (-1, 3))
        N = ro.shape[0]
        gd = None if gt_depth is None else self.arr(gt_depth, (N,))
        ga, k1 = self._grids(grids)
        da, k2 = self._decs(decoders)
        frag = np.zeros(N, self.dt)
        rc = self.lib.nso_ray_fragility(C.byref(opts), ga, da, _stage_id(stage), N, self._p(ro), self._p(rd),
                                        self._p(gd), self.creal(gt_depth_max), self._p(frag))
        assert rc == 0
        return frag

    # -- losses ----------------------------------------------------------------------------
    def loss_map(self, depth, rgb, gt_depth, gt_color, w_color, use_color):
        d, c, gd, gc = self.arr(depth), self.arr(rgb, (-1, 3)), self.arr(gt_depth), self.arr(gt_color, (-1, 3))
        N = d.shape[0]
        g_d, g_c = np.zeros(N, self.dt), np.zeros((N, 3), self.dt)
        loss = self.lib.nso_loss_map(N, self._p(d), self._p(c), self._p(gd), self._p(gc), self.creal(w_color),
                                     int(use_color), self._p(g_d), self._p(g_c))
        return float(loss), g_d, g_c

    def loss_track(self, depth, rgb, var, gt_depth, gt_color, w_color, use_color, handle_dynamic, detach_var=True):
        d, c, v = self.arr(depth), self.arr(rgb, (-1, 3)), self.arr(var)
        gd, gc = self.arr(gt_depth), self.arr(gt_color, (-1, 3))
        N = d.shape[0]
        g_d, g_c, g_v = np.zeros(N, self.dt), np.zeros((N, 3), self.dt), np.zeros(N, self.dt)
        loss = self.lib.nso_loss_track(N, self._p(d), self._p(c), self._p(v), self._p(gd), self._p(gc),
                                       self.creal(w_color), int(use_color), int(handle_dynamic), int(detach_var),
                                       self._p(g_d), self._p(g_c), self._p(g_v))
        return float(loss), g_d, g_c, g_v

    def adam_step(self, p, g, m, v, lr, step, mask=None, b1=0.9, b2=0.999, eps=1e-8):
        """in place on p, m, v (must be contiguous arrays of this oracle's dtype)"""
        for a in (p, m, v):
            assert a.dtype == self.dt and a.flags.c_contiguous
        g = self.arr(g)
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.lib.nso_adam_step(C.c_long(p.size), self._p(p), self._p(g), self._p(m), self._p(v), self._p(mk),
                               self.creal(lr), self.creal(b1), self.creal(b2), self.creal(eps), int(step))

    # -- pose / rays -----------------------------------------------------------------------
    def camera_from_tensor(self, cam):
        c = self.arr(cam, (7,))
        out = np.zeros((3, 4), self.dt)
        self.lib.nso_camera_from_tensor(self._p(c), self._p(out))
        return out

    def camera_backward(self, cam, g_c2w):
        c, g = self.arr(cam, (7,)), self.arr(g_c2w, (3, 4))
        out = np.zeros(7, self.dt)
        self.lib.nso_camera_backward(self._p(c), self._p(g), self._p(out))
        return out

    def rays_from_pixels(self, pix_i, pix_j, fx, fy, cx, cy, c2w, mode=0):
        pi = np.ascontiguousarray(pix_i, dtype=np.int32)
        pj = np.ascontiguousarray(pix_j, dtype=np.int32)
        n = pi.size
        m = self.arr(c2w)[:3, :4].copy()
        ro, rd = np.zeros((n, 3), self.dt), np.zeros((n, 3), self.dt)
        R = self.creal
        self.lib.nso_rays_from_pixels(n, self._p(pi), self._p(pj), R(fx), R(fy), R(cx), R(cy), self._p(m), mode,
                                      self._p(ro), self._p(rd))
        return ro, rd

    def rays_backward(self, pix_i, pix_j, fx, fy, cx, cy, g_rays_o, g_rays_d, mode=0):
        pi = np.ascontiguousarray(pix_i, dtype=np.int32)
        pj = np.ascontiguousarray(pix_j, dtype=np.int32)
        n = pi.size
        go, gd = self.arr(g_rays_o, (n, 3)), self.arr(g_rays_d, (n, 3))
        out = np.zeros((3, 4), self.dt)
        R = self.creal
        self.lib.nso_rays_backward(n, self._p(pi), self._p(pj), R(fx), R(fy), R(cx), R(cy), mode, self._p(go),
                                   self._p(gd), self._p(out))
        return out

    def world_to_camera(self, c2w):
        m = self.arr(c2w, (4, 4))
        out = np.zeros((4, 4), self.dt)
        self.lib.nso_world_to_camera(self._p(m), self._p(out))
        return out

    def frustum_mask(self, bound, shape_zyx, depth_img, intr, c2w, is_coarse=False):
        """Mapper::get_mask_from_c2w -> bool [Z,Y,X]"""
        b = self.arr(bound, (6,))
        d = self.arr(depth_img)
        H, W = d.shape
        Z, Y, X = shape_zyx
        m = self.arr(c2w, (4, 4))
        mask = np.zeros((Z, Y, X), np.uint8)
        R = self.creal
        fx, fy, cx, cy = intr
        self.lib.nso_frustum_mask(self._p(b), Z, Y, X, self._p(d), H, W, R(fx), R(fy), R(cx), R(cy), self._p(m), int(is_coarse), self._p(mask))
        return mask.astype(bool)

    def sample_pixels(self, seed, n, H0, H1, W0, W1):
        """raySampler's pixel draw (utils.h:19-36) with the library's hash -> (pix_i cols, pix_j rows) int32 [n]"""
        pi, pj = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self.lib.nso_sample_pixels(C.c_uint64(seed), n, H0, H1, W0, W1, self._p(pi), self._p(pj))
        return pi, pj

    def gather_pixels(self, pix_i, pix_j, depth_img, color_img):
        d, c = self.arr(depth_img), self.arr(color_img)
        pi, pj = np.ascontiguousarray(pix_i, np.int32), np.ascontiguousarray(pix_j, np.int32)
        gd, gc = np.zeros(len(pi), self.dt), np.zeros((len(pi), 3), self.dt)
        self.lib.nso_gather_pixels(len(pi), self._p(pi), self._p(pj), d.shape[1], self._p(d), self._p(c), self._p(gd), self._p(gc))
        return gd, gc

    def keyframe_overlap(self, rays_o, rays_d, gt_depth, intr, HW, c2w_list, n_samples=16):
        """Mapper::keyframe_selection_overlap: fraction of the current frame's sample points seen by each keyframe -> [K]"""
        ro, rd, gd = self.arr(rays_o, (-1, 3)), self.arr(rays_d, (-1, 3)), self.arr(gt_depth)
        m = self.arr(np.asarray(c2w_list).reshape(-1, 16))
        out = np.zeros(m.shape[0], self.dt)
        R = self.creal
        fx, fy, cx, cy = intr
        self.lib.nso_keyframe_overlap(ro.shape[0], self._p(ro), self._p(rd), self._p(gd), n_samples, int(HW[0]), int(HW[1]), R(fx), R(fy),
                                      R(cx), R(cy), m.shape[0], self._p(m), self._p(out))
        return out

    def inside_filter(self, bound, rays_o, rays_d, gt_depth):
        b = self.arr(bound, (6,))
        ro, rd, gd = self.arr(rays_o, (-1, 3)), self.arr(rays_d, (-1, 3)), self.arr(gt_depth)
        keep = np.zeros(ro.shape[0], np.uint8)
        self.lib.nso_inside_filter(self._p(b), ro.shape[0], self._p(ro), self._p(rd), self._p(gd), self._p(keep))
        return keep.astype(bool)
