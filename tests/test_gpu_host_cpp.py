"""GPU test (-m gpu) of the C++ host classes that keep the reference's surface (Renderer / NICE / Tracker / Mapper,
nice-slam-cpp_amd/host/): the driver host_test runs them through the C-ABI and dumps tensors; parity is judged here
against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import scenes
from scenes import rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "nice-slam-cpp_amd", "host", "host_test")


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    if not os.path.exists(EXE):
        pytest.fail("host_test is not built (run __graft_entry__.build())")
    d = tmp_path_factory.mktemp("host")
    r = subprocess.run([EXE, str(d)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return {f[:-4]: np.load(os.path.join(d, f)) for f in os.listdir(d) if f.endswith(".npy")}


def _scene(dump):
    grids = {k: dump["grid_" + k][0] for k in scenes.LEVELS}
    decs = {k: dump["dec_" + k] for k in scenes.LEVELS}
    return dump["bound"], grids, decs


def test_renderer_render_batch_ray_matches_oracle(dump, oracle32):
    bound, grids, decs = _scene(dump)
    op = oracle32.opts(bound)
    ref = oracle32.render_forward(op, grids, decs, "color", dump["rays_o"], dump["rays_d"], dump["gt_depth"])
    assert rel_l2(dump["r_depth"], ref["depth"]) < 1e-4
    assert rel_l2(dump["r_rgb"], ref["rgb"]) < 1e-4
    assert rel_l2(dump["r_var"], ref["var"]) < 1e-4
    assert rel_l2(dump["r_weights"], ref["weights"]) < 1e-4
    ref2 = oracle32.render_forward(op, grids, decs, "middle", dump["rays_o"], dump["rays_d"], None)
    assert dump["r2_weights"].shape == (150, 32)
    assert rel_l2(dump["r2_depth"], ref2["depth"]) < 1e-4 and rel_l2(dump["r2_weights"], ref2["weights"]) < 1e-4


def test_eval_points_and_nice_forward(dump, oracle32):
    bound, grids, decs = _scene(dump)
    pts = dump["pts"]
    # oracle raw through degenerate one-sample rays: p = o + d * z with z = near = 0.01 -> o = p - 0.01 d
    d = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (pts.shape[0], 1))
    o = (pts.astype(np.float64) - 0.01 * d).astype(np.float32)
    for stage, key in (("color", "raw"), ("fine", "raw_fine")):
        fw = oracle32.render_forward(oracle32.opts(bound, n_samples=1, n_surface=0), grids, decs, stage, o, d, None, want_aux=True)
        sel = np.abs(fw["z"][:, 0] - 0.01) < 1e-6
        got, ref = dump[key][sel], fw["raw"][sel, 0]
        assert sel.mean() > 0.9
        assert rel_l2(got, ref) < 2e-3            # p is reconstructed through fp32 o + d*z: embedding arguments differ by an ulp
        assert (got[:, 3] == 100).sum() == (ref[:, 3] == 100).sum()


def test_pose_helpers(dump, oracle32):
    assert rel_l2(dump["cam_RT"], oracle32.camera_from_tensor(dump["cam"])) < 1e-6


def test_tracker_moves_the_pose_and_reports_a_loss(dump):
    cam0, cam2 = dump["trk_cam0"], dump["trk_cam2"]
    step = np.abs(cam2 - cam0)
    assert np.isfinite(dump["trk_loss"]).all() and (dump["trk_loss"] > 0).all()
    assert (step > 1e-4).all() and (step < 2.5e-2).all()          # two Adam steps of lr 1e-2 on every component
    assert np.isfinite(dump["trk_run_cam"]).all() and dump["trk_run_cam"].shape == (7,)


def test_tracker_run_matches_oracle_on_the_same_pixel_draws(dump, oracle32, oracle64):
    """Tracker::run (src/Tracker.cpp:92-113) through the C++ class, every iteration resident on the device: pixel draw (counter-based
    hash, seed 1000 + i), ground-truth gather, rays from the pose, bound filter, render, loss with the dynamic-outlier median, backward
    onto the pose, Adam.  The oracle repeats the loop on the same draws; the pose after config `tracking.iters` = 3 iterations at
    `tracking.lr` = 1e-3 and the three losses must agree."""
    bound, grids, decs = _scene(dump)
    H, W, fx, fy, cx, cy = 48, 64, 40.0, 40.0, 32.0, 24.0
    depth_img, color_img, c2w = dump["map_depth_img"], dump["trk_color_img"], dump["map_c2w"]

    def run(o):
        cam = np.concatenate([[1.0, 0.0, 0.0, 0.0], c2w[:3, 3]]).astype(o.dt)         # get_tensor_from_camera of an identity rotation
        m, v = np.zeros(7, o.dt), np.zeros(7, o.dt)
        losses = []
        for i in range(3):
            pi, pj = o.sample_pixels(1000 + i, 100, 4, H - 4, 4, W - 4)
            gd, gc = o.gather_pixels(pi, pj, depth_img, color_img)
            ro, rd = o.rays_from_pixels(pi, pj, fx, fy, cx, cy, o.camera_from_tensor(cam))
            keep = o.inside_filter(bound, ro, rd, gd)
            op = o.opts(bound)
            fw = o.render_forward(op, grids, decs, "color", ro[keep], rd[keep], gd[keep])
            l, gD, gC, gV = o.loss_track(fw["depth"], fw["rgb"], fw["var"], gd[keep], gc[keep], 0.5, True, True, True)
            bw = o.render_backward(op, grids, decs, "color", ro[keep], rd[keep], gd[keep], -1.0, gC, gD, None, want_grids=False, want_decoders=False)
            g = o.camera_backward(cam, o.rays_backward(pi[keep], pj[keep], fx, fy, cx, cy, bw["g_rays_o"], bw["g_rays_d"]))
            o.adam_step(cam, g, m, v, 1e-3, i + 1)
            losses.append(l)
        return cam, losses
    cam32, l32 = run(oracle32)
    cam64, _ = run(oracle64)
    got = dump["trk_run_cam"]
    assert np.abs(cam32[4:] - c2w[:3, 3]).max() > 1e-3                                  # the three steps moved the pose
    for a, r in zip(dump["trk_run_losses"], l32):
        assert abs(a - r) < 1e-3 * abs(r), (a, r)
    e, e64, eo = rel_l2(got, cam32), rel_l2(got, cam64), rel_l2(cam32, cam64)
    assert e < 1e-4 or e64 < 2 * eo + 1e-4, (e, e64, eo)


def test_a_new_frame_is_uploaded_whatever_addresses_its_tensors_have(dump):
    """the device-resident frame cache (DevFrame, keyed by the host tensors): frame B run after frame A on the same Tracker -- A's tensors freed, B's
    the same size, so the allocator may hand out A's addresses -- must give exactly what a Tracker that never saw A gives, and not A's result"""
    loss_a, loss_b_after_a, loss_b_fresh = [float(x) for x in dump["frame_identity"]]
    assert loss_b_after_a == loss_b_fresh and loss_b_after_a != loss_a, (loss_a, loss_b_after_a, loss_b_fresh)


def test_thin_mapper_methods_of_the_reference_surface(dump, oracle32):
    """Mapper::get_mask_from_c2w(cv::Mat, ...) and Mapper::keyframe_selection_overlap(...) (include/Mapper.h:24-25) as thin methods"""
    bound, grids, decs = _scene(dump)
    fm = oracle32.frustum_mask(bound, grids["middle"].shape[1:], dump["map_depth_img"], (40.0, 40.0, 32.0, 24.0), dump["map_c2w"])     # [Z,Y,X]
    got = dump["thin_mask_middle_xyz"] > 0.5                                            # [X,Y,Z] like the reference's result
    assert got.shape == fm.shape[::-1] and np.array_equal(got.transpose(2, 1, 0), fm)
    assert [int(x) for x in dump["thin_selected"]] == [1]                               # keyframe 1 sees the frame, keyframe 0 looks away (dropped: 0 overlap)
    assert dump["thin_overlap"][0] == 0 and dump["thin_overlap"][1] > 0.005            # (20-pixel edge of a 64x48 image: a small window)


def test_mapper_optimises_grids_and_colour_decoder(dump, oracle32):
    bound, grids, decs = _scene(dump)
    assert np.isfinite(dump["map_loss"]).all()
    for k in ("middle", "fine", "color"):
        new = dump["map_grid_" + k][0]
        assert np.isfinite(new).all() and np.abs(new - grids[k]).max() > 1e-4, k
    mask = dump["map_fine_mask"] > 0.5
    assert np.array_equal(dump["map_grid_fine"][0][:, ~mask], grids["fine"][:, ~mask])      # frustum mask: unmarked voxels never move
    # the other levels carry the device frustum mask of the current frame (Mapper.cpp:231-281 -> nsk_frustum_mask)
    for k in ("middle", "color"):
        fm = oracle32.frustum_mask(bound, grids[k].shape[1:], dump["map_depth_img"], (40.0, 40.0, 32.0, 24.0), dump["map_c2w"])
        new = dump["map_grid_" + k][0]
        assert 0 < fm.sum() < fm.size
        assert np.array_equal(new[:, ~fm], grids[k][:, ~fm]) and np.abs(new[:, fm] - grids[k][:, fm]).max() > 1e-4, k
    assert float(dump["map_dec_color_delta"][0]) > 1e-5 and float(dump["map_dec_fine_delta"][0]) == 0.0   # fix_fine, !fix_color
    # the Renderer picks the optimised state up without explicit uploads: compare with the oracle on the dumped state
    g2 = dict(grids)
    for k in ("middle", "fine", "color"):
        g2[k] = dump["map_grid_" + k][0]
    d2 = dict(decs); d2["color"] = dump["dec_color_after"]
    ref = oracle32.render_forward(oracle32.opts(bound), g2, d2, "color", dump["rays_o"], dump["rays_d"], dump["gt_depth"])
    assert rel_l2(dump["r3_depth"], ref["depth"]) < 1e-4 and rel_l2(dump["r3_rgb"], ref["rgb"]) < 1e-4


def _tensor_from_camera(c2w):
    """get_tensor_from_camera (utils.h:212-231 as the host shim restates it): (w, x, y, z, tx, ty, tz), Shepperd's method in double"""
    R = np.asarray(c2w, np.float64)[:3, :3]
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2; q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2; q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2; q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2; q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    return np.concatenate([np.asarray(q, np.float32), np.asarray(c2w, np.float32)[:3, 3]]).astype(np.float32)


def _replay_mapper_run(o, dump):
    """Mapper::run / optimize_map (src/Mapper.cpp:198-552 as nice-slam-cpp_amd/host/src/nsk_host.cpp restates it) on the oracle: six frames,
    the same counter-hash pixel draws (Mapper::draw_seed), stage schedule, lr_factor, overlap window, frustum masks, pixs_per_image and
    bundle adjustment; returns per-iteration losses, windows, final grids, colour decoder, frame poses and keyframe poses"""
    M64 = (1 << 64) - 1
    H, W, fx, fy, cx, cy = 48, 64, 40.0, 40.0, 32.0, 24.0
    intr = (fx, fy, cx, cy)
    bound = dump["bound"]
    dt = o.dt
    grids = {k: dump["mo_grid_%s_0" % k][0].astype(dt).copy() for k in scenes.LEVELS}
    decs = {k: dump["mo_dec_" + k].astype(dt).copy() for k in ("coarse", "middle", "fine")}
    decs["color"] = dump["mo_dec_color_0"].astype(dt).copy()
    depth_img, color_img = dump["map_depth_img"], dump["trk_color_img"]
    est = [m.astype(np.float32).copy() for m in dump["mo_poses_0"]]
    # config of host_test's scenario (NS_YAML with the edits there)
    iters_first, iters, lr_first, lr_later, window_size, pixels = 4, 3, 2.0, 1.0, 5, 200
    mid_ratio, fine_ratio, w_color, ba_lr = np.float32(0.4), np.float32(0.6), 0.5, 1e-3
    stage_lr = {"middle": dict(dec=0.0, middle=0.1, fine=0.0, color=0.0), "fine": dict(dec=0.0, middle=0.005, fine=0.005, color=0.0),
                "color": dict(dec=0.005, middle=0.005, fine=0.005, color=0.005)}
    seed, calls = 4321, 0
    draw = lambda d: (seed + 0xD1B54A32D192ED03 * d) & M64
    kfs, kf_idx = [], []                         # keyframes: est_c2w (mutable by BA)
    losses, windows = [], []
    first = True
    for idx in range(6):
        lr_factor, n_it = (lr_first, iters_first) if first else (lr_later, iters)
        cur = est[idx].copy()
        BA = len(kf_idx) > 4
        # ---- optimize_map ----
        nkf = len(kfs)
        win = []
        if nkf > 1:                              # keyframe_selection_overlap over keyframes[:-1]
            calls += 1
            K = nkf - 1
            pi, pj = o.sample_pixels((draw(calls) + 0x9e3779b9 * (K + 1)) & M64, 100, 0, H, 0, W)
            gd, _ = o.gather_pixels(pi, pj, depth_img, color_img)
            ro, rd = o.rays_from_pixels(pi, pj, *intr, cur)
            pct = o.keyframe_overlap(ro, rd, gd, intr, (H, W), [k for k in kfs[:-1]])
            order = sorted([k for k in range(K) if pct[k] > 0], key=lambda k: -pct[k])      # (stable: ties keep index order)
            win = order[:window_size - 2]
        if nkf > 0:
            win.append(nkf - 1)
        oldest = min(win) if win else -1
        win.append(-1)
        windows.append(list(win))
        nf = len(win)
        per = pixels // nf
        masks = {k: o.frustum_mask(bound, grids[k].shape[1:], depth_img, intr, cur, is_coarse=(k == "coarse")) for k in ("middle", "fine", "color")}
        mom = {k: (np.zeros_like(grids[k]), np.zeros_like(grids[k])) for k in ("middle", "fine", "color")}
        dm, dv = np.zeros_like(decs["color"]), np.zeros_like(decs["color"])
        pose_of = lambda f: (kfs[f] if f != -1 else cur)
        is_ba = [BA and f != oldest for f in win]
        cams = [(_tensor_from_camera(pose_of(f)).astype(dt) if b else None) for f, b in zip(win, is_ba)]
        cm = [np.zeros(7, dt) for _ in win]; cv = [np.zeros(7, dt) for _ in win]
        calls += 1
        call_seed = draw(calls)
        ba_step = 0
        it_losses = []
        ba_grads = np.zeros((nf, 7), dt)
        op = o.opts(bound)
        for it in range(n_it):
            stage = "middle" if it <= int(np.float32(n_it) * mid_ratio) else ("fine" if it <= int(np.float32(n_it) * fine_ratio) else "color")
            ba_now = any(is_ba) and stage == "color"
            pis, pjs, ros, rds, gds, gcs, keeps = [], [], [], [], [], [], []
            for i, f in enumerate(win):
                pi, pj = o.sample_pixels((call_seed + 0x100000001b3 * (it * nf + i + 1)) & M64, per, 0, H, 0, W)
                gd, gc = o.gather_pixels(pi, pj, depth_img, color_img)           # (every frame of this scenario carries the same images)
                c2w = o.camera_from_tensor(cams[i]) if is_ba[i] else np.asarray(pose_of(f), dt)[:3, :4]
                ro, rd = o.rays_from_pixels(pi, pj, *intr, c2w)
                keeps.append(o.inside_filter(bound, ro, rd, gd))
                pis.append(pi); pjs.append(pj); ros.append(ro); rds.append(rd); gds.append(gd); gcs.append(gc)
            keep = np.concatenate(keeps)
            ro, rd, gd, gc = np.concatenate(ros)[keep], np.concatenate(rds)[keep], np.concatenate(gds)[keep], np.concatenate(gcs)[keep]
            fw = o.render_forward(op, grids, decs, "color", ro, rd, gd)          # :430 renders the literal "color" (D19)
            loss, g_d, g_c = o.loss_map(fw["depth"], fw["rgb"], gd, gc, w_color, stage == "color")
            it_losses.append(loss)
            bw = o.render_backward(op, grids, decs, "color", ro, rd, gd, -1.0, g_c, g_d, None, want_rays=ba_now)
            L = stage_lr[stage]
            for k in ("middle", "fine", "color"):
                o.adam_step(grids[k], bw["g_grids"][k], mom[k][0], mom[k][1], L[k] * lr_factor, it + 1, mask=np.repeat(masks[k][None], 32, 0))
            o.adam_step(decs["color"], bw["g_decoders"]["color"], dm, dv, L["dec"] * lr_factor, it + 1)
            if ba_now:
                ba_step += 1
                g_ro = np.zeros((per * nf, 3), dt); g_rd = np.zeros((per * nf, 3), dt)
                g_ro[keep] = bw["g_rays_o"]; g_rd[keep] = bw["g_rays_d"]
                for i in range(nf):
                    if not is_ba[i]:
                        continue
                    sl = slice(i * per, (i + 1) * per)
                    g_cam = o.camera_backward(cams[i], o.rays_backward(pis[i], pjs[i], *intr, g_ro[sl], g_rd[sl]))
                    ba_grads[i] = g_cam
                    o.adam_step(cams[i], g_cam, cm[i], cv[i], ba_lr, ba_step)
        losses.append(it_losses)
        if any(is_ba):
            for i, f in enumerate(win):
                if not is_ba[i]:
                    continue
                c2w = np.eye(4, dtype=np.float32); c2w[:3, :4] = o.camera_from_tensor(cams[i])
                if f != -1:
                    kfs[f] = c2w
                else:
                    cur = c2w
        # ---- rest of run() ----
        if BA:
            est[idx] = cur
        if idx not in kf_idx:                    # keyframe_every: 1
            kf_idx.append(idx); kfs.append(cur.copy())
        first = False
    return losses, windows, grids, decs["color"], np.stack(est), np.stack(kfs), ba_grads, win


def test_mapper_run_matches_oracle_on_the_same_pixel_draws(dump, oracle32, oracle64):
    """Mapper::run through the C++ class (six frames, all keyframes; the sixth optimises with bundle adjustment) against the oracle replaying
    the same loop on the same pixel draws: windows exactly; the loss of every iteration; the optimised grids (free run: elements whose
    Adam step is decided by a rounding-sized gradient are counted, the rest must agree), colour decoder, and the poses bundle adjustment
    rewrote.  The fp64 replay arbitrates what two fp32 evaluations of this loop can differ by.  Ref: src/Mapper.cpp:198-491,493-552."""
    l32, w32, g32, d32, est32, kf32, bg32, win = _replay_mapper_run(oracle32, dump)
    l64, w64, g64, d64, est64, kf64, bg64, _ = _replay_mapper_run(oracle64, dump)
    got_w = [[int(x) for x in row if x > -8.5] for row in dump["mo_windows"]]
    assert got_w == w32 == w64, (got_w, w32)
    assert got_w[5][-1] == -1 and len(got_w[5]) == 5                                         # the full window of mapping_window_size frames
    got_l = [[float(x) for x in row if x >= 0] for row in dump["mo_losses"]]
    assert [len(r) for r in got_l] == [4, 3, 3, 3, 3, 3]
    worst_l = 0.0
    for a, b, c in zip(got_l, l32, l64):
        for x, y, z in zip(a, b, c):
            worst_l = max(worst_l, abs(x - y) / abs(y))
            assert abs(x - y) < max(2e-3 * abs(y), 3 * abs(y - z)), (got_l, l32)
    # bundle adjustment ran on the last frame: its pose and the keyframe poses moved, and agree with the replay
    est1, kf1 = dump["mo_poses_1"], dump["mo_kf_poses_1"]
    assert np.abs(est1[5] - dump["mo_poses_0"][5]).max() > 1e-5 and np.abs(kf1[:5] - dump["mo_poses_0"][:5]).max() > 1e-5
    assert np.array_equal(est1[:5], dump["mo_poses_0"][:5])                                  # frames before BA keep their given poses
    # Bundle adjustment: the pose gradient sums, over 40 rays x 48 samples per frame, terms of size |B| ~ 25..75 times the decoder gradient with
    # both signs -- with these random decoders a handful of ReLU flips moves it by tens of percent (the fp32 and fp64 ORACLES differ by 3 % on
    # it).  Until round 3 the GPU's gradient therefore agreed with the fp32 oracle's "in direction only" (cos 0.86 .. 0.99); since round 4 both see
    # the same sample points and embedding arguments bit for bit and take (all but a handful of) the same branches: the gradients must agree to
    # 1e-2 relative and cos > 0.9999 (measured: 2.7e-3 at worst, cos 1.000 on all four frames) -- and the step is still teacher-forced: the oracle's Adam and quad2rotation applied to the GPU's own gradient
    # must give the GPU's poses.
    bg = dump["mo_ba_grad"]
    assert bg.shape == bg32.shape and [int(f) for f in got_w[5]] == list(win)
    cosines, nba = [], 0
    o = oracle32
    for i, f in enumerate(win):
        start = dump["mo_poses_0"][5] if f == -1 else dump["mo_poses_0"][f]
        pose_gpu = (est1[5] if f == -1 else kf1[f])
        if not np.any(bg32[i]):
            assert not np.any(bg[i]) and np.array_equal(pose_gpu, start)                     # the oldest frame of the window stays fixed (:305-329)
            continue
        nba += 1
        cosines.append(float(np.dot(bg[i], bg32[i]) / (np.linalg.norm(bg[i]) * np.linalg.norm(bg32[i]))))
        cam = _tensor_from_camera(start).astype(np.float32)
        m, v = np.zeros(7, np.float32), np.zeros(7, np.float32)
        o.adam_step(cam, bg[i].astype(np.float32), m, v, 1e-3, 1)
        ref = o.camera_from_tensor(cam)
        assert np.abs(pose_gpu[:3] - ref).max() < 2e-6, (i, f, pose_gpu[:3], ref)
    assert nba == 4 and min(cosines) > 0.9999, cosines
    for i in range(len(win)):
        if np.any(bg32[i]):
            assert rel_l2(bg[i], bg32[i]) < 1e-2, (i, rel_l2(bg[i], bg32[i]))
    e_pose, eo_pose = rel_l2(est1[5][:3], est32[5][:3]), rel_l2(est32[5][:3], est64[5][:3])
    flips = {}
    for k in ("middle", "fine", "color"):
        got = dump["mo_grid_%s_1" % k][0]
        assert np.abs(got - dump["mo_grid_%s_0" % k][0]).max() > 1e-3, k
        far = np.abs(got - g32[k]) > 0.1 * 0.005                                            # an Adam step of the other sign (lr 0.005 .. 0.2)
        far64 = np.abs(g32[k].astype(np.float64) - g64[k]) > 0.1 * 0.005
        # (19 iterations at learning rates up to 0.2 on grids of std 0.3: one open sign early on moves a voxel by 0.4 and the runs part there --
        # the losses above, which every schedule / learning-rate / mask / window error would move at once, are the sharp check of the loop)
        flips[k] = (round(float(far.mean()), 4), round(float(far64.mean()), 4), float(rel_l2(got[~far], g32[k][~far])))
        assert far.mean() < max(0.12, 3 * far64.mean()), (k, flips[k])
        assert flips[k][2] < 1e-3, (k, flips[k])
    e_dec, eo_dec = rel_l2(dump["mo_dec_color_1"], d32), rel_l2(d32, d64)
    assert np.abs(dump["mo_dec_color_1"] - dump["mo_dec_color_0"]).max() > 1e-4
    assert e_dec < max(2e-3, 3 * eo_dec), (e_dec, eo_dec)
    print("Mapper::run vs oracle: losses <= %.1e, BA pose gradients cos %s (teacher-forced step exact), BA pose %.1e (fp32 vs fp64 oracle %.1e), "
          "colour decoder %.1e (%.1e), grid elements with the other Adam sign %s" % (worst_l, ["%.3f" % c for c in cosines], e_pose, eo_pose, e_dec, eo_dec, flips))


def test_mapper_window_is_ranked_by_overlap(dump, oracle32):
    """Mapper::keyframe_selection_overlap (src/Mapper.cpp:132-216) through the C++ Mapper: the window of the sixth frame holds
    the mapping_window_size-2 = 3 best-overlapping keyframes of [0..3] (fractions from nsk_keyframe_overlap on 100 random
    pixels, ranked on the host: descending, zero overlap dropped), then the last keyframe (4), then the current frame (-1).
    The fractions themselves are checked against the oracle on a regular pixel lattice of the same frame (sampling noise
    allowed); their exact agreement on identical rays is test_gpu_parity.py::test_keyframe_overlap_matches_oracle."""
    poses, pct = dump["kf_poses"], dump["kf_overlap"]
    assert pct.shape == (4,)
    order = [int(k) for k in np.argsort(-pct, kind="stable") if pct[k] > 0][:3]
    assert [int(x) for x in dump["kf_window"]] == order + [4, -1], (dump["kf_window"], pct)
    H, W, fx, fy, cx, cy = 48, 64, 40.0, 40.0, 32.0, 24.0
    jj, ii = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    cur = poses[5].astype(np.float64)
    dirs = np.stack([(ii - cx) / fx, -(jj - cy) / fy, -np.ones_like(ii, dtype=np.float64)], -1).reshape(-1, 3)
    rd = dirs @ cur[:3, :3].T
    ro = np.broadcast_to(cur[:3, 3], rd.shape)
    ref = oracle32.keyframe_overlap(ro, rd, dump["map_depth_img"].reshape(-1), (fx, fy, cx, cy), (H, W), poses[:4])
    assert ref[1] == 0 and pct[1] == 0 and ref[0] > ref[2] > ref[3] > 0        # opposite view never overlaps
    assert np.abs(pct - ref).max() < 0.08, (pct, ref)


NS_YAML = """coarse: True
grid_len:
  coarse: 2.0
  middle: 0.32
  fine: 0.16
  color: 0.16
tracking:
  ignore_edge_W: 4
  ignore_edge_H: 4
  use_color_in_tracking: True
  handle_dynamic: True
  w_color_loss: 0.5
  lr: 0.001
  pixels: 200
  iters: 5
mapping:
  color_refine: True
  middle_iter_ratio: 0.4
  fine_iter_ratio: 0.6
  BA: False
  BA_cam_lr: 0.001
  fix_fine: True
  fix_color: False
  keyframe_every: 2
  mapping_window_size: 5
  w_color_loss: 0.2
  frustum_feature_selection: True
  keyframe_selection_method: 'overlap'
  lr_first_factor: 5
  lr_factor: 1
  pixels: 500
  iters_first: 30
  iters: 10
  stage:
    coarse:
      decoders_lr: 0.0
      coarse_lr: 0.001
      middle_lr: 0.0
      fine_lr: 0.0
      color_lr: 0.0
    middle:
      decoders_lr: 0.0
      coarse_lr: 0.0
      middle_lr: 0.1
      fine_lr: 0.0
      color_lr: 0.0
    fine:
      decoders_lr: 0.0
      coarse_lr: 0.0
      middle_lr: 0.005
      fine_lr: 0.005
      color_lr: 0.0
    color:
      decoders_lr: 0.005
      coarse_lr: 0.0
      middle_lr: 0.005
      fine_lr: 0.005
      color_lr: 0.005
"""


def test_slam_loop_over_a_tum_sequence(tmp_path):
    """K5's frame loop through the C++ classes: SequenceReader (TUM layout) -> Tracker::run on every frame -> Mapper::run (keyframes,
    overlap window, frustum masks) on every frame, five frames of a synthetic sequence rendered from the analytic room and written as
    16-bit depth / 8-bit colour PNGs with TUM stamp files.  The decoders are random (no pretrained weights exist offline), so this
    checks that the loop runs end to end on the device and stays sane -- finite losses, a falling mapping loss on the first frame's
    30 iterations vs its later ones is not asserted -- poses that stay within centimetres of the truth the Tracker was started near."""
    from test_host_io import _png
    exe = os.path.join(ROOT, "nice-slam-cpp_amd", "host", "slam_loop")
    if not os.path.exists(exe):
        pytest.fail("slam_loop is not built (run __graft_entry__.build())")
    seq = tmp_path / "tum"
    (seq / "rgb").mkdir(parents=True); (seq / "depth").mkdir()
    H, W, fx, fy, cx, cy = 48, 64, 40.0, 40.0, 32.0, 24.0
    bound = np.array([[-2.0, 2.0], [-1.5, 1.5], [-2.0, 2.2]], np.float32)
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    gts = []
    with open(seq / "rgb.txt", "w") as fr, open(seq / "depth.txt", "w") as fd, open(seq / "groundtruth.txt", "w") as fg:
        for i in range(5):
            t = 1.0 + 0.1 * i
            p_tum = np.eye(4); p_tum[:3, 3] = [0.02 * i, 0.01 * i, 0.015 * i]          # TUM convention; frame 0 = identity (poses are relative to it)
            c2w = (p_tum @ flip).astype(np.float32)                                   # the OpenGL camera the reader hands out
            gts.append(c2w)
            depth = scenes.frame_depth_image(bound, c2w, H, W, fx, fy, cx, cy)
            jj, ii = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
            dirs = np.stack([(ii - cx) / fx, -(jj - cy) / fy, -np.ones_like(ii, dtype=np.float64)], -1) @ c2w[:3, :3].T.astype(np.float64)
            hit = c2w[:3, 3] + dirs * depth[..., None]
            col = np.clip(255 * (0.5 + 0.5 * np.sin(hit * np.array([1.3, 2.1, 0.7]))), 0, 255).astype(np.uint8)
            _png(str(seq / "rgb" / ("%.6f.png" % t)), col)
            _png(str(seq / "depth" / ("%.6f.png" % t)), np.round(depth * 5000).astype(np.uint16))
            fr.write("%.6f rgb/%.6f.png\n" % (t, t)); fd.write("%.6f depth/%.6f.png\n" % (t, t))
            fg.write("%.6f %.9g %.9g %.9g 0 0 0 1\n" % ((t,) + tuple(p_tum[:3, 3])))
    (seq / "bound.txt").write_text(" ".join("%g" % v for v in bound.reshape(-1)))
    ns, cf = tmp_path / "ns.yaml", tmp_path / "cf.yaml"
    ns.write_text(NS_YAML)
    cf.write_text("mapping:\n  pixels: 500\ncam:\n  H: %d\n  W: %d\n  fx: %g\n  fy: %g\n  cx: %g\n  cy: %g\n" % (H, W, fx, fy, cx, cy))
    out = tmp_path / "out"; out.mkdir()
    r = subprocess.run([exe, "tum", str(seq), str(ns), str(cf), str(out), "5", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "slam_loop ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    est, gt = np.load(out / "est_poses.npy"), np.load(out / "gt_poses.npy")
    assert est.shape == (5, 4, 4) and np.allclose(gt, np.stack(gts), atol=1e-5)                     # the reader's poses are the ones written
    tl, ml = np.load(out / "track_loss.npy"), np.load(out / "map_loss.npy")
    assert tl.shape == (5,) and ml.shape == (5,) and np.isfinite(tl).all() and np.isfinite(ml).all() and (ml > 0).all() and (tl[1:] > 0).all()
    assert np.isfinite(est).all()
    assert np.linalg.norm(est[:, :3, 3] - gt[:, :3, 3], axis=1).max() < 0.15                        # started from the previous estimate, 5 small steps
    for k in ("grid_middle", "grid_fine", "grid_color"):
        g = np.load(out / (k + ".npy"))
        assert np.isfinite(g).all() and np.abs(g).max() > 0.02                                     # the map moved away from its N(0, 0.01) init


def test_slam_loop_at_the_reference_config_sizes(tmp_path, capsys):
    """K5 at the sizes config/nice_slam.yaml states, through the C++ classes: 640 x 480 frames (TUM fr1 intrinsics), Tracker 200 pixels x 10
    iterations on every frame, Mapper 1000 pixels x 1500 iterations on the first frame and x 60 on every fifth, window 5, frustum feature
    selection, BA on (it starts with the fifth keyframe: not reached in six frames).  The decoders are random and the oracle cannot walk
    1 560 iterations of 48 000 samples in test time, so beyond "runs, finite, the pose stays near the truth, the map moved" this records what
    the loop costs on the device: the Tracker's ten iterations and the Mapper's mean iteration (printed)."""
    from test_host_io import _png
    exe = os.path.join(ROOT, "nice-slam-cpp_amd", "host", "slam_loop")
    if not os.path.exists(exe):
        pytest.fail("slam_loop is not built (run __graft_entry__.build())")
    seq = tmp_path / "tum"
    (seq / "rgb").mkdir(parents=True); (seq / "depth").mkdir()
    H, W, fx, fy, cx, cy = 480, 640, 517.3, 516.5, 318.6, 255.3
    bound = np.array([[-2.0, 2.0], [-1.5, 1.5], [-2.0, 2.2]], np.float32)
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    F = 6
    gts = []
    with open(seq / "rgb.txt", "w") as fr, open(seq / "depth.txt", "w") as fd, open(seq / "groundtruth.txt", "w") as fg:
        for i in range(F):
            t = 1.0 + 0.1 * i
            p_tum = np.eye(4); p_tum[:3, 3] = [0.01 * i, 0.005 * i, 0.008 * i]
            c2w = (p_tum @ flip).astype(np.float32)
            gts.append(c2w)
            depth = scenes.frame_depth_image(bound, c2w, H, W, fx, fy, cx, cy)
            jj, ii = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
            dirs = np.stack([(ii - cx) / fx, -(jj - cy) / fy, -np.ones_like(ii, dtype=np.float64)], -1) @ c2w[:3, :3].T.astype(np.float64)
            hit = c2w[:3, 3] + dirs * depth[..., None]
            col = np.clip(255 * (0.5 + 0.5 * np.sin(hit * np.array([1.3, 2.1, 0.7]))), 0, 255).astype(np.uint8)
            _png(str(seq / "rgb" / ("%.6f.png" % t)), col)
            _png(str(seq / "depth" / ("%.6f.png" % t)), np.round(depth * 5000).astype(np.uint16))
            fr.write("%.6f rgb/%.6f.png\n" % (t, t)); fd.write("%.6f depth/%.6f.png\n" % (t, t))
            fg.write("%.6f %.9g %.9g %.9g 0 0 0 1\n" % ((t,) + tuple(p_tum[:3, 3])))
    (seq / "bound.txt").write_text(" ".join("%g" % v for v in bound.reshape(-1)))
    ns, cf = tmp_path / "ns.yaml", tmp_path / "cf.yaml"
    y = NS_YAML
    for a, b in (("  ignore_edge_W: 4", "  ignore_edge_W: 20"), ("  ignore_edge_H: 4", "  ignore_edge_H: 20"), ("  pixels: 200\n  iters: 5", "  pixels: 200\n  iters: 10"),
                 ("  BA: False", "  BA: True"), ("  keyframe_every: 2", "  keyframe_every: 50"), ("  pixels: 500\n  iters_first: 30\n  iters: 10", "  pixels: 1000\n  iters_first: 1500\n  iters: 60")):
        assert y.count(a) == 1, a
        y = y.replace(a, b)
    ns.write_text(y)
    cf.write_text("mapping:\n  pixels: 1000\ncam:\n  H: %d\n  W: %d\n  fx: %g\n  fy: %g\n  cx: %g\n  cy: %g\n" % (H, W, fx, fy, cx, cy))
    out = tmp_path / "out"; out.mkdir()
    r = subprocess.run([exe, "tum", str(seq), str(ns), str(cf), str(out), str(F), "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "slam_loop ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    times = [l for l in r.stdout.splitlines() if l.startswith("time frame")]
    with capsys.disabled():
        print("\nK5 at the reference config's sizes (640 x 480, Tracker 200 x 10, Mapper 1000 x 1500 / 60):")
        for l in times:
            print("  " + l)
    trk = [float(l.split("tracker loop")[1].split("us")[0]) for l in times if "tracker loop" in l]
    mp = [float(l.split("mapper iteration")[1].split("us")[0]) for l in times if "mapper iteration" in l]
    assert len(trk) == F - 1 and len(mp) == 2
    assert max(trk[1:]) < 5000 and max(mp) < 2000                                                   # ten Tracker iterations < 5 ms, a Mapper iteration < 2 ms
    est, gt = np.load(out / "est_poses.npy"), np.load(out / "gt_poses.npy")
    assert est.shape == (F, 4, 4) and np.allclose(gt, np.stack(gts), atol=1e-5)
    tl, ml = np.load(out / "track_loss.npy"), np.load(out / "map_loss.npy")
    assert tl.shape == (F,) and ml.shape == (2,) and np.isfinite(tl).all() and np.isfinite(ml).all() and (ml > 0).all() and (tl[1:] > 0).all()
    assert np.isfinite(est).all() and np.linalg.norm(est[:, :3, 3] - gt[:, :3, 3], axis=1).max() < 0.15
    for k in ("grid_middle", "grid_fine", "grid_color"):
        g = np.load(out / (k + ".npy"))
        assert np.isfinite(g).all() and np.abs(g).max() > 0.02


def test_sharded_mapper_with_bundle_adjustment_equals_the_single_process_run(tmp_path):
    """BASELINE configs[4] at N > 1 through the C++ class (Mapper::set_distributed): six frames of Mapper::run, every one a
    keyframe, the sixth with bundle adjustment, with the window's rays sharded over THREE ranks -- one process per rank as on a node, all three
    on this box's one GPU, the exchange a rank-ordered sum through shared memory (RCCL refuses several ranks on one device; the RCCL form of the
    same call is nsk_allreduce_grads, tests/test_gpu_dist.py).  203 pixels per iteration: the shards are uneven and window frames split across
    ranks.  Every rank draws the whole batch (the pixel draw is a hash of the seed), renders its contiguous shard with the batch's max(gt_depth)
    taken over the whole batch on the device (nsk_set_depth_max_batch), and ONE exchange per iteration carries the marked voxels' gradients, the
    colour decoder's, the loss and -- in the BA iterations -- 8 floats of pose gradient per window frame (nsk_grad_extra); grids, decoder and poses
    then take the same Adam steps on every rank.  Checked: the ranks end bit-identical; their losses, bundle-adjustment pose gradients, poses and
    parameters equal the single-process run's up to the order of the sums; the shards' kept rays add up to the batch's."""
    exe = os.path.join(ROOT, "nice-slam-cpp_amd", "host", "dist_test")
    if not os.path.exists(exe):
        pytest.fail("dist_test is not built (run __graft_entry__.build())")
    d = str(tmp_path)
    r = subprocess.run([exe, d, "0", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    single = {f[3:-4]: np.load(os.path.join(d, f)) for f in os.listdir(d) if f.startswith("r0_")}
    world = 3
    d2 = os.path.join(d, "w3"); os.makedirs(d2)
    shm = "/dev/shm/nsk_dist_test_%d" % os.getpid()
    with open(shm, "wb") as f:
        f.truncate(4096 + world * (4 << 20) * 4)
    try:
        procs = [subprocess.Popen([exe, d2, str(k), str(world), shm], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for k in range(world)]
        outs = [p.communicate(timeout=400)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    finally:
        os.remove(shm)
    ranks = [{f[3:-4]: np.load(os.path.join(d2, f)) for f in os.listdir(d2) if f.startswith("r%d_" % k)} for k in range(world)]
    for k in range(1, world):                                    # replicas stay in step without any broadcast
        for name in ranks[0]:
            assert np.array_equal(ranks[0][name], ranks[k][name]), (k, name)
    got = ranks[0]
    # the Tracker ran on rank 0 only and its pose went out as a sum with zeros (Dist::broadcast0): every rank holds rank 0's pose (the loop above
    # compared the files bit for bit), and it is the pose the single process tracked, up to the order of its ray-gradient sums
    assert got["tracked"].shape == (8,) and got["tracked"][7] == 0 and abs(np.linalg.norm(got["tracked"][:4]) - 1) < 1e-2
    assert np.abs(got["tracked"] - single["tracked"]).max() < 1e-4, (got["tracked"], single["tracked"])
    assert np.allclose(got["losses"], single["losses"], rtol=2e-5, atol=1e-6), (got["losses"], single["losses"])
    assert got["kept"][0] > 0 and single["ba_grad"].shape == got["ba_grad"].shape
    # bundle-adjustment pose gradients: sums of the same ray gradients in another order
    assert rel_l2(got["ba_grad"], single["ba_grad"]) < 1e-3, (got["ba_grad"], single["ba_grad"])
    assert np.abs(got["poses"] - single["poses"]).max() < 1e-5 and np.abs(got["kf_poses"] - single["kf_poses"]).max() < 1e-5
    assert np.abs(got["kf_poses"] - got["kf_poses"][:1]).max() > 0                      # (the poses are not all one pose: BA moved some)
    for name in ("grid_middle", "grid_fine", "grid_color", "dec_color"):
        far = np.abs(got[name] - single[name]) > 2e-3                                      # an Adam step on a rounding-sized gradient goes either way
        assert far.mean() < 1e-2, (name, far.mean())
        assert rel_l2(got[name][~far], single[name][~far]) < 2e-3, name
    print("sharded Mapper, 3 ranks: losses within %.1e of the single-process run, BA pose gradients %.1e, poses %.1e; kept rays of the last BA iteration %d" % (
        np.abs(got["losses"] - single["losses"]).max(), rel_l2(got["ba_grad"], single["ba_grad"]), np.abs(got["poses"] - single["poses"]).max(), int(got["kept"][0])))
