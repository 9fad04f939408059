"""helpers shared by the -m gpu tests"""
import numpy as np
import torch

import scenes


def make_ctx(sc, n_samples=32, n_surface=16, occupancy=False, trainable=()):
    import nice_slam_cpp_amd as pkg
    ctx = pkg.Context(0)
    ctx.set_render_opts(n_samples=n_samples, n_surface=n_surface, occupancy=occupancy)
    ctx.load_scene(sc["bound"], sc["grids"], sc["decoders"])
    for k in trainable:
        ctx.decoder_set_trainable(k, True)
    return ctx


def cu(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def stage_levels(stage):
    return {"coarse": ["coarse"], "middle": ["middle"], "fine": ["middle", "fine"], "color": ["middle", "fine", "color"]}[stage]


def robust_rel_l2(a, b):
    return scenes.rel_l2(a, b)
