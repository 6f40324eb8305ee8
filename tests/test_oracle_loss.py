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


def test_mapper_oracle_weight_matches_reference_weighting(gold):
    from oracle import mapper_oracle as mo
    w = mo.weight(gold["wg_cur"].astype(np.float64), gold["wg_glob"].astype(np.float64))[:, None]
    for k in ("_xyz", "_rgb", "_scaling", "_rotation", "_opacity"):
        np.testing.assert_allclose(gold["wg_in_" + k] * w, gold["wg_out_" + k], rtol=1e-6, atol=0)


def test_mapper_oracle_activation_chain_rule_by_finite_differences():
    from oracle import mapper_oracle as mo
    rng = np.random.default_rng(3)
    n = 6
    ro, rs, rq = rng.normal(size=(n, 1)), rng.normal(size=(n, 2)) * 0.3, rng.normal(size=(n, 4))
    co, cs, cq = rng.normal(size=(n, 1)), rng.normal(size=(n, 2)), rng.normal(size=(n, 4))

    def f(ro, rs, rq):
        o, s, q = mo.activate(ro, rs, rq)
        return float((o * co).sum() + (s * cs).sum() + (q * cq).sum())

    o, s, q = mo.activate(ro, rs, rq)
    cur = np.stack([np.ones(n), np.zeros(n)], 1)           # weight = 1 / (1 + 1e-6 + 1)
    d = {"means3D": np.zeros((n, 3)), "colors": np.zeros((n, 3)), "opacities": co, "scales": cs,
         "rotations": cq}
    out, vis, *_ = mo.grad_finish(o, s, rq, d, cur, np.ones(n, np.int32), np.zeros(n, bool),
                                  np.zeros((n, 2)), np.zeros((n, 2)), np.zeros(n),
                                  np.zeros(n, np.int64), 4)
    wgt = 1.0 / (2.0 + 1e-6)
    eps = 1e-6
    for name, arr in (("_opacity", ro), ("_scaling", rs), ("_rotation", rq)):
        num = np.zeros_like(arr)
        for idx in np.ndindex(arr.shape):
            a = arr.copy(); a[idx] += eps
            b = arr.copy(); b[idx] -= eps
            args_p = [a if arr is x else x for x in (ro, rs, rq)]
            args_m = [b if arr is x else x for x in (ro, rs, rq)]
            num[idx] = (f(*args_p) - f(*args_m)) / (2 * eps)
        np.testing.assert_allclose(out[name], num * wgt, rtol=1e-6, atol=1e-8, err_msg=name)
    assert vis.all()


def test_postprocess_guarded_depth_matches_nan_to_num():
    allmap = torch.rand(9, 7, 9, dtype=lo.DT)
    allmap[1, 2, 3] = 0.0          # x / 0 -> inf -> 0
    allmap[0, 4, 4] = 0.0
    allmap[1, 4, 4] = 0.0          # 0 / 0 -> nan -> 0
    intr = {"fu": 5.0, "fv": 5.0, "cu": 3.0, "cv": 4.0}
    out = lo.postprocess(torch.zeros(3, 7, 9, dtype=lo.DT), allmap, torch.eye(4, dtype=lo.DT), intr)
    want = torch.nan_to_num(allmap[0:1] / allmap[1:2], 0, 0, 0)
    assert torch.equal(out["depth"], want)
