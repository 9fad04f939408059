"""Deterministic synthetic scenes for parity tests and the bench: an analytic box
room inside the reference's hard-coded bound, seeded grids and decoder weights, rays from seeded pixels.
Pure numpy so that it runs unchanged on the GPU box (no reference files, no datasets)."""
import numpy as np

REF_BOUND = np.array([[-4.5, 3.82], [-1.5, 2.02], [-3.0, 2.76]], dtype=np.float32)   # src/main.cpp:33
# grid shapes of src/main.cpp:38-75 evaluated in fp32 for the bound above
REF_GRID_SHAPES = {"coarse": (32, 5, 3, 8), "middle": (32, 18, 11, 26), "fine": (32, 36, 22, 52),
                   "color": (32, 36, 22, 52)}
SMALL_GRID_SHAPES = {"coarse": (32, 3, 2, 4), "middle": (32, 6, 5, 7), "fine": (32, 9, 8, 11), "color": (32, 9, 8, 11)}
# Scenes the reference has no configuration for (BASELINE.json configs[2..4]): bounds and intrinsics are
# DECLARED HERE, in the harness -- ScanNet-scene0000-class room (K3), Replica-office0-class room (K4), TUM-fr1/desk-class volume (K5)
K3_BOUND = np.array([[0.0, 8.6], [0.0, 8.9], [-0.3, 3.3]], dtype=np.float32)
K4_BOUND = np.array([[-2.2, 3.8], [-1.7, 3.3], [-1.4, 1.6]], dtype=np.float32)
K5_BOUND = np.array([[-3.5, 3.0], [-3.0, 3.0], [-3.0, 3.0]], dtype=np.float32)
CAM_NICE_SLAM = dict(H=680, W=1200, fx=600.0, fy=600.0, cx=599.5, cy=339.5)          # config/nice_slam.yaml:96-102 (K2, K4)
CAM_SCANNET = dict(H=480, W=640, fx=577.590698, fy=578.729797, cx=318.905426, cy=242.683609)   # K3
CAM_TUM = dict(H=480, W=640, fx=517.3, fy=516.5, cx=318.6, cy=255.3)                 # K5
UP = {"K2": "y", "K3": "z", "K4": "z", "K5": "z"}                                     # which world axis is the height in each scene
GRID_LEN = {"coarse": 2.0, "middle": 0.32, "fine": 0.16, "color": 0.16}              # config/nice_slam.yaml:7-11
E_DIM, H_DIM = 93, 32
LEVELS = ("coarse", "middle", "fine", "color")


def grid_shapes_for(bound, c_dim=32, coarse_bound_enlarge=2):
    """grid shapes [C,Z,Y,X] of src/main.cpp:34-75 for a bound (fp32 arithmetic, truncation by .item<int>())"""
    b = np.asarray(bound, np.float32)
    xyz = (b[:, 1] - b[:, 0]).astype(np.float32)
    out = {}
    for k in LEVELS:
        v = xyz * np.float32(coarse_bound_enlarge) / np.float32(GRID_LEN[k]) if k == "coarse" else xyz / np.float32(GRID_LEN[k])
        out[k] = (c_dim, int(v[2]), int(v[1]), int(v[0]))
    return out


def decoder_param_count(which):
    c_dim = 64 if which == "fine" else 32
    out_dim = 4 if which == "color" else 1
    if which == "coarse":
        return 3 * (32 * 32 + 32) + (64 * 32 + 32) + (32 * 32 + 32) + out_dim * 32 + out_dim
    n = 3 * E_DIM + (E_DIM * 32 + 32) + 3 * (32 * 32 + 32) + ((32 + E_DIM) * 32 + 32)
    return n + 5 * (c_dim * 32 + 32) + out_dim * 32 + out_dim


def make_decoder(which, rng, bias_std=0.0):
    """packed parameters: B~N(0,25^2) (GaussianFFT.cpp:6), xavier-uniform(gain sqrt2) weights, zero bias
    (MLP.cpp:65-74), fc = nn.Linear default init.  bias_std>0 perturbs all biases so bias paths are tested."""
    has_xyz = which != "coarse"
    c_dim = 64 if which == "fine" else 32
    out_dim = 4 if which == "color" else 1
    in_dims = [E_DIM, 32, 32, 32 + E_DIM, 32] if has_xyz else [32, 32, 32, 64, 32]
    parts = []
    if has_xyz:
        parts.append(rng.standard_normal((3, E_DIM)) * 25.0)
    for fi in in_dims:
        a = np.sqrt(2.0) * np.sqrt(6.0 / (fi + 32))
        parts.append(rng.uniform(-a, a, (32, fi)))
        parts.append(rng.standard_normal(32) * bias_std)
    if has_xyz:
        for _ in range(5):
            k = 1.0 / np.sqrt(c_dim)
            parts.append(rng.uniform(-k, k, (32, c_dim)))
            parts.append(rng.uniform(-k, k, 32))
    a = np.sqrt(2.0) * np.sqrt(6.0 / (32 + out_dim))
    parts.append(rng.uniform(-a, a, (out_dim, 32)))
    parts.append(rng.standard_normal(out_dim) * bias_std)
    P = np.concatenate([p.reshape(-1) for p in parts]).astype(np.float32)
    assert P.size == decoder_param_count(which)
    return P


def make_scene(seed=0, grid_shapes=None, bound=None, grid_std=None, bias_std=0.0):
    """grids [C,Z,Y,X] float32 ~N(0,std) (src/main.cpp:44,54,65,76: 0.01, fine 1e-4) + packed decoders"""
    rng = np.random.default_rng(seed)
    shapes = grid_shapes or REF_GRID_SHAPES
    std = {"coarse": 0.01, "middle": 0.01, "fine": 1e-4, "color": 0.01}
    if grid_std is not None:
        std = {k: grid_std for k in std}
    grids = {k: (rng.standard_normal(shapes[k]) * std[k]).astype(np.float32) for k in LEVELS}
    decoders = {k: make_decoder(k, rng, bias_std) for k in LEVELS}
    return dict(bound=(REF_BOUND if bound is None else np.asarray(bound, np.float32)).copy(), grids=grids,
                decoders=decoders)


def _ray_box_far(box, o, d):
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (box[None, :, :] - o[:, :, None]) / d[:, :, None]
    return np.min(np.max(t, axis=2), axis=1)


def look_rotation(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return Ry @ Rx @ Rz


def make_camera(rng, bound, up="y"):
    """a pose well inside the room looking roughly horizontally (OpenGL camera, utils.h:47: it looks along its own -z).  up = "y": the
    world's y axis is the height (the reference's hard-coded bound, src/main.cpp:33); up = "z": z is the height (ScanNet / Replica /
    TUM conventions: K3-K5), so the camera frame is first turned to look along world +y with its +y along world +z"""
    ctr = bound.mean(axis=1)
    ext = bound[:, 1] - bound[:, 0]
    t = ctr + (rng.uniform(-0.15, 0.15, 3) * ext)
    R = look_rotation(rng.uniform(-0.6, 0.6), rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2))
    if up == "z":
        R = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]]) @ R
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, :3] = R
    c2w[:3, 3] = t
    return c2w


def frame_depth_image(bound, c2w, H, W, fx, fy, cx, cy, shrink=0.3):
    """z-depth image [H,W] of the analytic room (bound shrunk by `shrink`) seen from c2w: what a depth camera of make_rays' scene records"""
    room = np.asarray(bound, np.float64).copy()
    room[:, 0] += shrink
    room[:, 1] -= shrink
    jj, ii = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    dirs = np.stack([(ii - cx) / fx, -(jj - cy) / fy, -np.ones_like(ii, dtype=np.float64)], -1).reshape(-1, 3) @ c2w[:3, :3].T.astype(np.float64)
    o = np.broadcast_to(c2w[:3, 3].astype(np.float64), dirs.shape)
    return _ray_box_far(room, o, dirs).reshape(H, W).astype(np.float32)


def frame_color_image(bound, c2w, H, W, fx, fy, cx, cy, shrink=0.3):
    """colour image [H,W,3] of the same room: make_rays' smooth function of the hit point, for every pixel"""
    room = np.asarray(bound, np.float64).copy()
    room[:, 0] += shrink
    room[:, 1] -= shrink
    jj, ii = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    dirs = np.stack([(ii - cx) / fx, -(jj - cy) / fy, -np.ones_like(ii, dtype=np.float64)], -1).reshape(-1, 3) @ c2w[:3, :3].T.astype(np.float64)
    o = np.broadcast_to(c2w[:3, 3].astype(np.float64), dirs.shape)
    hit = o + dirs * _ray_box_far(room, o, dirs)[:, None]
    return (0.5 + 0.5 * np.sin(hit * np.array([1.3, 2.1, 0.7]) + np.array([0.0, 1.0, 2.0]))).reshape(H, W, 3).astype(np.float32)


def make_rays(seed, n, bound, H=680, W=1200, fx=600.0, fy=600.0, cx=599.5, cy=339.5, n_frames=1, zero_frac=0.05,
              shrink=0.3, edge=0, cam_seed=None, up="y"):
    """rays of n seeded pixels split over n_frames seeded cameras (Mapper.cpp:223: pixels/|window| each);
    gt depth = z-depth of the ray/room intersection (room = bound shrunk by `shrink`), zero_frac of them
    set to 0 (Renderer.cpp:94-98 branch); gt colour = smooth function of the hit point.  Intrinsics default
    to config/nice_slam.yaml:97-102."""
    rng = np.random.default_rng(seed)
    crng = rng if cam_seed is None else np.random.default_rng(cam_seed)      # cam_seed: the same window of cameras for every pixel draw
    bound = np.asarray(bound, np.float32)
    room = bound.astype(np.float64).copy()
    room[:, 0] += shrink
    room[:, 1] -= shrink
    per = n // n_frames
    ro, rd, pi, pj, fr, cams = [], [], [], [], [], []
    for f in range(n_frames):
        c2w = make_camera(crng, bound, up)
        cams.append(c2w)
        i = rng.integers(edge, W - edge, per)
        j = rng.integers(edge, H - edge, per)
        dirs = np.stack([(i - cx) / fx, -(j - cy) / fy, -np.ones(per)], -1)
        rd.append((dirs @ c2w[:3, :3].T.astype(np.float64)))
        ro.append(np.broadcast_to(c2w[:3, 3].astype(np.float64), (per, 3)))
        pi.append(i); pj.append(j); fr.append(np.full(per, f))
    ro, rd = np.concatenate(ro), np.concatenate(rd)
    depth = _ray_box_far(room, ro, rd)
    hit = ro + rd * depth[:, None]
    color = 0.5 + 0.5 * np.sin(hit * np.array([1.3, 2.1, 0.7]) + np.array([0.0, 1.0, 2.0]))
    zero = rng.random(ro.shape[0]) < zero_frac
    depth = np.where(zero, 0.0, depth)
    c32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(rays_o=c32(ro), rays_d=c32(rd), gt_depth=c32(depth), gt_color=c32(color), pix_i=np.concatenate(pi).astype(np.int32),
                pix_j=np.concatenate(pj).astype(np.int32), frame=np.concatenate(fr).astype(np.int32),
                c2w=np.stack(cams).astype(np.float32), intr=(fx, fy, cx, cy), HW=(H, W))


def rel_l2(a, b):
    """relative L2 error of a against reference b (north_star tolerance metric)"""
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    den = np.linalg.norm(b)
    if den == 0:
        return float(np.linalg.norm(a))
    return float(np.linalg.norm(a - b) / den)
