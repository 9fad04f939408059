"""experiment: what the packed exchange costs a K3 step on ONE GPU without the collective (pack + unpack launches only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
W = bench.workloads()
for xchg in (False, True, False, True):
    import nice_slam_cpp_amd as pkg, scenes
    wl = W["K3"]
    sc = scenes.make_scene(42, scenes.grid_shapes_for(wl["bound"]), bound=wl["bound"])
    res = None
    # reuse bench's runner by monkey-patching its world-size check: run_workload(..., world=1) never exchanges, so time by hand
    ctx = pkg.Context(0); ctx.set_render_opts(); ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"]); ctx.decoder_set_trainable("color", True)
    cam = wl["cam"]
    rays = scenes.make_rays(1234, wl["rays"], sc["bound"], H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], n_frames=1, cam_seed=4242, up=wl["up"])
    depth = scenes.frame_depth_image(sc["bound"], rays["c2w"][0], cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    cu = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
    d_img = cu(depth)
    for k in ("middle", "fine", "color"):
        ctx.frustum_mask(k, d_img, (cam["fx"], cam["fy"], cam["cx"], cam["cy"]), rays["c2w"][0])
    t = [cu(rays[k]) for k in ("rays_o", "rays_d", "gt_depth", "gt_color")]
    loss = torch.zeros(1, device="cuda")
    lr = bench.STAGE_LR["color"]
    def step():
        ctx.map_step("color", *t, float(rays["gt_depth"].max()), 0.2, True, flags=3, loss=loss)
        if xchg:
            ctx.grad_pack(); ctx.grad_unpack()
        ctx.adam_step(lr)
    with torch.cuda.stream(ctx.tstream):
        for _ in range(20): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): step()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pack+unpack %s: %.4f ms/step" % (xchg, dt / 200 * 1e3), flush=True)
    ctx.close()
