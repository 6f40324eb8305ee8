"""The float64 loss oracle (oracle/loss_oracle.py) against golden vectors produced by the
reference's own get_loss / ssim / depth->normal functions
(tests/golden/make_harness_golden.py).  Runs on CPU."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as lo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _t(a):
    return torch.from_numpy(np.asarray(a)).to(lo.DT)


def test_total_loss_and_gradients(gold):
    pred = {k: _t(gold["loss_in_" + k]).requires_grad_(True)
            for k in ("rgb", "accum", "depth", "normal", "surf_normal", "dist")}
    gt = {"rgb": _t(gold["loss_gt_rgb"]), "depth": _t(gold["loss_gt_depth"]),
          "depth_cov": _t(gold["loss_gt_cov"])}
    w = {"rgb_loss": 1.0, "depth_loss": 1.0, "alpha_loss": 1.0, "normal_loss": 0.1, "dist_loss": 0.05}
    loss = lo.total_loss(pred, gt, w)
    loss.backward()
    # the golden values are float32 results of the reference: float32 tolerances
    np.testing.assert_allclose(loss.item(), float(gold["loss_value"]), rtol=2e-6)
    for k in pred:
        np.testing.assert_allclose(pred[k].grad.numpy(), gold["loss_grad_" + k], rtol=3e-4, atol=2e-9,
                                   err_msg=k)


def test_ssim_map(gold):
    got = lo.ssim_map(_t(gold["loss_in_rgb"])[None], _t(gold["loss_gt_rgb"])[None])[0]
    # the golden map is a float32 result (variance by cancellation): absolute tolerance
    np.testing.assert_allclose(got.numpy(), gold["ssim_map"], rtol=1e-5, atol=2e-5)


def test_depth_to_normal(gold):
    fu, fv, cu, cv = (float(x) for x in gold["normal_intr"])
    intr = {"fu": fu, "fv": fv, "cu": cu, "cv": cv}
    got = lo.depth_to_normal(_t(gold["normal_depth"]), intr)
    np.testing.assert_allclose(got.numpy(), gold["normal_out"], rtol=1e-5, atol=1e-6)


def test_postprocess_guarded_depth_matches_nan_to_num():
    allmap = torch.rand(9, 7, 9, dtype=lo.DT)
    allmap[1, 2, 3] = 0.0          # x / 0 -> inf -> 0
    allmap[0, 4, 4] = 0.0
    allmap[1, 4, 4] = 0.0          # 0 / 0 -> nan -> 0
    intr = {"fu": 5.0, "fv": 5.0, "cu": 3.0, "cv": 4.0}
    out = lo.postprocess(torch.zeros(3, 7, 9, dtype=lo.DT), allmap, torch.eye(4, dtype=lo.DT), intr)
    want = torch.nan_to_num(allmap[0:1] / allmap[1:2], 0, 0, 0)
    assert torch.equal(out["depth"], want)
