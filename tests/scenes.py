"""Seeded scene builders shared by the oracle tests and the GPU parity tests."""
import numpy as np
import torch

from vings_mono_amd import synth


def scene(n, W, H, seed=synth.SEED_BASE, view_k=0, n_views=1, scale_mul=1.0, z_range=(1.5, 60.0),
          with_flow=False, bg=(0.0, 0.0, 0.0), kind="frustum"):
    """Activated splats + camera matrices as float32 numpy arrays.  kind = "surfaces": opaque
    splats on a ground plane, walls and facades (synth.make_surface_splats)."""
    if kind == "surfaces":
        raw = synth.make_surface_splats(n, W, H, n_views=n_views, seed=seed)
    else:
        raw = synth.make_splats(n, W, H, n_views=n_views, seed=seed, z_range=z_range)
    act = synth.activate(raw)
    c2w = synth.window_poses(max(n_views, view_k + 1))[view_k]
    w2c = torch.linalg.inv(c2w)
    cam = synth.Camera(w2c, synth.intrinsic_for(W, H))
    out = {k: v.numpy().astype(np.float32).copy() for k, v in act.items()}
    out["scales"] = out["scales"] * np.float32(scale_mul)
    out["view"] = cam.world_view_transform.numpy().reshape(16).copy()
    out["proj"] = cam.full_proj_transform.numpy().reshape(16).copy()
    out["bg"] = np.asarray(bg, np.float32)
    out["tanfovx"], out["tanfovy"] = cam.tanfovx, cam.tanfovy
    out["campos"] = cam.camera_center.numpy().copy()
    out["H"], out["W"] = H, W
    if with_flow:
        out["flow"] = np.random.default_rng(seed).normal(size=(n, 2)).astype(np.float32)
    else:
        out["flow"] = None
    return out


def oracle_forward(orc, sc, pixel_mask=None):
    dt = orc.dtype
    return orc.forward(sc["means3D"].astype(dt), sc["scales"].astype(dt), sc["rotations"].astype(dt),
                       sc["opacities"].astype(dt), sc["colors"].astype(dt), sc["view"].astype(dt),
                       sc["proj"].astype(dt), sc["bg"].astype(dt), sc["H"], sc["W"],
                       pixel_mask=pixel_mask,
                       flow=None if sc["flow"] is None else sc["flow"].astype(dt))
