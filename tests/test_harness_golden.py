"""The Python wrapper around the rasteriser (camera matrices, loss, depth->normal,
score weighting) against golden vectors produced by the reference's own
functions (tests/golden/make_harness_golden.py imports them from
/root/reference in the development container).  Runs on CPU.
"""
import math
import os

import numpy as np
import pytest
import torch

from vings_mono_amd import harness, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_camera_matrices_bit_exact(gold):
    for row in gold["cameras"]:
        W, H, fx, fy, fovx, fovy, tanx, tany = row[:8]
        P_ref = row[8:].reshape(4, 4)
        assert synth.focal_to_fov(fx, W) == fovx and synth.focal_to_fov(fy, H) == fovy
        P = synth.projection_matrix(fovx, fovy).numpy().astype(np.float64)
        np.testing.assert_array_equal(P, P_ref)
        cam = synth.Camera(torch.eye(4), {"fu": fy, "fv": fx, "cu": 0, "cv": 0, "H": int(H), "W": int(W)})
        assert cam.tanfovx == tanx and cam.tanfovy == tany
        np.testing.assert_array_equal(cam.projection_matrix.numpy().astype(np.float64), P_ref.T)


def test_camera_transposed_layout():
    w2c = torch.tensor([[0.0, -1, 0, 0.3], [1, 0, 0, -0.2], [0, 0, 1, 1.5], [0, 0, 0, 1]])
    cam = synth.Camera(w2c, synth.intrinsic_for(64, 48))
    np.testing.assert_array_equal(cam.world_view_transform.numpy(), w2c.T.numpy())
    np.testing.assert_allclose(cam.full_proj_transform.numpy(),
                               (w2c.T @ cam.projection_matrix).numpy())
    np.testing.assert_allclose(cam.camera_center.numpy(), torch.linalg.inv(w2c)[:3, 3].numpy(), atol=1e-6)


def test_loss_value_and_gradients(gold):
    pred = {k: torch.from_numpy(gold["loss_in_" + k]).clone().requires_grad_(True)
            for k in ("rgb", "accum", "depth", "normal", "surf_normal", "dist")}
    gt = {"rgb": torch.from_numpy(gold["loss_gt_rgb"]), "depth": torch.from_numpy(gold["loss_gt_depth"]),
          "depth_cov": torch.from_numpy(gold["loss_gt_cov"])}
    w = {"rgb_loss": 1.0, "depth_loss": 1.0, "alpha_loss": 1.0, "normal_loss": 0.1, "dist_loss": 0.05}
    loss = harness.mapping_loss(pred, gt, w)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(gold["loss_value"]), rtol=2e-6)
    for k in pred:
        np.testing.assert_allclose(pred[k].grad.numpy(), gold["loss_grad_" + k], rtol=2e-5, atol=2e-9,
                                   err_msg=k)


def test_ssim_map(gold):
    got = harness.ssim_map(torch.from_numpy(gold["loss_in_rgb"])[None],
                           torch.from_numpy(gold["loss_gt_rgb"])[None])[0]
    np.testing.assert_allclose(got.numpy(), gold["ssim_map"], rtol=1e-5, atol=1e-6)


def test_depth_to_normal(gold):
    fu, fv, cu, cv = gold["normal_intr"]
    H, W = gold["normal_depth"].shape
    intr = {"fu": float(fu), "fv": float(fv), "cu": float(cu), "cv": float(cv), "H": H, "W": W}
    got = harness.depth_to_normal(torch.from_numpy(gold["normal_depth"]), intr)
    np.testing.assert_allclose(got.numpy(), gold["normal_out"], rtol=1e-5, atol=1e-6)


def test_score_weighting(gold):
    w = harness.score_weight(torch.from_numpy(gold["wg_cur"]), torch.from_numpy(gold["wg_glob"]))
    for k in ("_xyz", "_rgb", "_scaling", "_rotation", "_opacity"):
        np.testing.assert_allclose((torch.from_numpy(gold["wg_in_" + k]) * w).numpy(),
                                   gold["wg_out_" + k], rtol=1e-6, atol=0)


def test_postprocess_depth_and_normal_rotation():
    """depth = allmap0/allmap1 with nan->0; normals rotate by w2c[:3,:3] as row vectors
    (/root/reference/scripts/gaussian/gaussian_base.py:205-221)."""
    H, W = 6, 8
    allmap = torch.zeros(9, H, W)
    allmap[0] = 2.0
    allmap[1] = 0.5
    allmap[1, 0, 0] = 0.0
    allmap[0, 0, 0] = 0.0
    allmap[2:5] = torch.tensor([0.0, 0.0, 1.0]).view(3, 1, 1)
    a = math.radians(30)
    w2c = torch.eye(4)
    w2c[:3, :3] = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    out = harness.postprocess(torch.zeros(3, H, W), torch.zeros(4), allmap, w2c,
                              synth.intrinsic_for(W, H))
    assert out["depth"][0, 0, 0] == 0 and torch.all(out["depth"][0, 1:, 1:] == 4.0)
    np.testing.assert_allclose(out["normal"][:, 2, 2].numpy(), w2c[2, :3].numpy(), atol=1e-6)
