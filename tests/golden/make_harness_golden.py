"""Generates tests/golden/harness_golden.npz by IMPORTING the reference's own
Python helpers (run in the development container only; /root/reference does not
exist on the GPU box).  The fixture holds inputs and expected outputs only.

Reference functions exercised (all pure PyTorch / math, importable without the
absent native extensions):
  gaussian.cameras.focal2fov, getProjectionMatrix      cameras.py:7-28
  gaussian.loss_utils.get_loss (+ autograd gradients)   loss_utils.py:110-153
  gaussian.normal_utils.depth_propagate_normal + TFer   normal_utils.py:6-21, tf.py:133-142
  gaussian.gaussian_utils.weighting_grad                gaussian_utils.py:56-62
  gaussian.gaussian_utils.distCUDA2                     gaussian_utils.py:12-16

Usage:  python tests/golden/make_harness_golden.py
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference/scripts"
sys.path.insert(0, REF)

from gaussian import cameras as ref_cam            # noqa: E402
from gaussian import loss_utils as ref_loss        # noqa: E402
from gaussian import normal_utils as ref_normal    # noqa: E402
from gaussian import gaussian_utils as ref_gutils  # noqa: E402
from gaussian.tf import TFer                       # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness_golden.npz")


def main():
    g = torch.Generator().manual_seed(20250124)
    out = {}
    # ---- cameras -------------------------------------------------------------
    cams = []
    for (W, H, f) in [(640, 480, 384.0), (1920, 1080, 1152.0), (616, 344, 300.0), (728, 264, 552.5)]:
        fovx, fovy = ref_cam.focal2fov(f, W), ref_cam.focal2fov(f * 1.01, H)
        P = ref_cam.getProjectionMatrix(fovX=fovx, fovY=fovy)
        cams.append(np.concatenate([[W, H, f, f * 1.01, fovx, fovy, math.tan(fovx * 0.5),
                                     math.tan(fovy * 0.5)], P.numpy().reshape(-1).astype(np.float64)]))
    out["cameras"] = np.stack(cams)
    # ---- loss ----------------------------------------------------------------
    H, W = 40, 56
    gt_rgb = torch.rand(3, H, W, generator=g)
    gt_rgb[:, :6] = 0.0                      # sky rows
    gt_depth = torch.rand(1, H, W, generator=g) * 5 + 1
    gt_depth[:, :, :4] = 0.0                 # invalid depth columns
    cov = torch.rand(1, H, W, generator=g) * 0.5 + 0.5
    pred = {"rgb": torch.rand(3, H, W, generator=g), "accum": torch.rand(1, H, W, generator=g),
            "depth": torch.rand(1, H, W, generator=g) * 5 + 1,
            "normal": torch.randn(3, H, W, generator=g), "surf_normal": torch.randn(3, H, W, generator=g),
            "dist": torch.rand(1, H, W, generator=g)}
    for k in pred:
        pred[k].requires_grad_(True)
    cfg = {"use_sky": False, "training_args": {"loss_weights": {
        "rgb_loss": 1.0, "depth_loss": 1.0, "alpha_loss": 1.0, "normal_loss": 0.1, "dist_loss": 0.05}}}
    gt = {"rgb": gt_rgb, "depth": gt_depth, "depth_cov": cov}
    loss = ref_loss.get_loss(cfg, pred, gt)
    loss.backward()
    out["loss_value"] = np.array(loss.item())
    for k in pred:
        out["loss_in_" + k] = pred[k].detach().numpy()
        out["loss_grad_" + k] = pred[k].grad.numpy()
    out["loss_gt_rgb"], out["loss_gt_depth"], out["loss_gt_cov"] = gt_rgb.numpy(), gt_depth.numpy(), cov.numpy()
    out["ssim_map"] = ref_loss.ssim_img(pred["rgb"].detach()[None], gt_rgb[None])[0].numpy()
    # ---- depth -> normal -------------------------------------------------------
    intr = {"fu": 52.0, "fv": 50.0, "cu": 19.3, "cv": 27.7, "H": H, "W": W}
    tfer = TFer({"intrinsic": intr, "device": {"mapper": "cpu"}})
    depth = torch.rand(H, W, generator=g) * 4 + 1
    out["normal_depth"] = depth.numpy()
    out["normal_intr"] = np.array([intr["fu"], intr["fv"], intr["cu"], intr["cv"]])
    out["normal_out"] = ref_normal.depth_propagate_normal(depth, tfer).numpy()
    # ---- score weighting ---------------------------------------------------------
    class M:  # the function only touches .grad of five tensors
        pass
    m = M()
    N = 257
    shapes = {"_xyz": 3, "_rgb": 3, "_scaling": 2, "_rotation": 4, "_opacity": 1}
    for k, c in shapes.items():
        t = torch.zeros(N, c)
        t.grad = torch.randn(N, c, generator=g)
        out["wg_in_" + k] = t.grad.numpy().copy()
        setattr(m, k, t)
    cur = torch.rand(N, 2, generator=g)
    glob = torch.rand(N, 2, generator=g) * 3
    cur[::7, 0] = 0.0
    ref_gutils.weighting_grad(m, cur, glob)
    out["wg_cur"], out["wg_glob"] = cur.numpy(), glob.numpy()
    for k in shapes:
        out["wg_out_" + k] = getattr(m, k).grad.numpy()
    # ---- mean squared distance to the 3 nearest neighbours (scale of new splats) ---------
    pts = torch.randn(3000, 3, generator=g) * torch.tensor([4.0, 2.0, 8.0])
    pts[10] = pts[11]                        # a duplicated point: its zero distance counts
    out["knn_points"] = pts.numpy()
    out["knn_dist2"] = ref_gutils.distCUDA2(pts).numpy()
    # ---- point cloud of a new keyframe (get_pointcloud, gaussian_utils.py:19-54) ----------
    Hp, Wp = 36, 52
    pintr = {"fu": 41.0, "fv": 43.5, "cu": 17.2, "cv": 26.1, "H": Hp, "W": Wp}
    ptfer = TFer({"intrinsic": pintr, "device": {"mapper": "cpu"}, "adc_args": {"accum_thresh": 0.6}})
    pdepth = torch.rand(1, Hp, Wp, generator=g) * 6 + 0.5
    pdepth[:, :, :5] = 0.0
    pdepth[:, 30:, 40:] = 0.0
    prgb = torch.rand(3, Hp, Wp, generator=g)
    paccum = torch.rand(1, Hp, Wp, generator=g)
    ang = 0.3
    pc2w = torch.tensor([[math.cos(ang), 0.0, math.sin(ang), 1.5], [0.0, 1.0, 0.0, -0.4],
                         [-math.sin(ang), 0.0, math.cos(ang), 2.0], [0.0, 0.0, 0.0, 1.0]])
    torch.manual_seed(77)
    xyz, col, q = ref_gutils.get_pointcloud(ptfer, pc2w, prgb, pdepth, paccum, 900)
    torch.manual_seed(77)
    perm = torch.randperm(int((pdepth > 0).sum()))
    out["pc_intr"] = np.array([pintr["fu"], pintr["fv"], pintr["cu"], pintr["cv"]])
    out["pc_depth"], out["pc_rgb"], out["pc_accum"] = pdepth.numpy(), prgb.numpy(), paccum.numpy()
    out["pc_c2w"], out["pc_perm"] = pc2w.numpy(), perm.numpy()
    out["pc_xyz"], out["pc_col"] = xyz.numpy(), col.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
