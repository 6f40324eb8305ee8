"""CPU oracle for the fused mapping-loss kernels: float64 PyTorch restatement of the
reference's per-view loss graph, differentiated by autograd.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; it is used by
tests/ to check csrc/mapping_loss.hip.  Pinned against golden vectors generated from the
reference's own functions (tests/golden/harness_golden.npz, made by
tests/golden/make_harness_golden.py): see tests/test_oracle_loss.py.

Follows
  /root/reference/scripts/gaussian/gaussian_base.py:205-224  render post-processing
  /root/reference/scripts/gaussian/normal_utils.py:6-21      depth -> normal
  /root/reference/scripts/gaussian/tf.py:133-142             back-projection (u = row)
  /root/reference/scripts/gaussian/loss_utils.py:26-57       SSIM
  /root/reference/scripts/gaussian/loss_utils.py:110-153     get_loss, use_sky = False
with one deliberate difference shared with the product: an empty mask divides by 1
instead of yielding NaN.
"""
import math

import torch
import torch.nn.functional as F

DT = torch.float64


def points_from_depth(depth, intr):
    H, W = depth.shape
    rows = torch.arange(H, dtype=depth.dtype).view(H, 1)
    cols = torch.arange(W, dtype=depth.dtype).view(1, W)
    return torch.stack(((cols - intr["cv"]) / intr["fv"] * depth,
                        (rows - intr["cu"]) / intr["fu"] * depth, depth), dim=-1)


def depth_to_normal(depth, intr):
    pts = points_from_depth(depth, intr)
    right_minus_left = pts[1:-1, 2:] - pts[1:-1, :-2]
    up_minus_down = pts[:-2, 1:-1] - pts[2:, 1:-1]
    n = F.normalize(torch.cross(right_minus_left, up_minus_down, dim=-1), p=2, dim=-1)
    return F.pad(n, (0, 0, 1, 1, 1, 1))


def postprocess(rgb, allmap, w2c, intr):
    alpha = allmap[1:2]
    # same values as nan_to_num(allmap0 / alpha, 0, 0); written with a guarded denominator so
    # that autograd returns 0 instead of NaN (0 * inf) at the pixels it replaces
    with torch.no_grad():
        ok = torch.isfinite(allmap[0:1] / alpha)
    depth = torch.where(ok, allmap[0:1] / torch.where(ok, alpha, torch.ones_like(alpha)),
                        torch.zeros_like(alpha))
    R = w2c[:3, :3].to(allmap.dtype)
    normal = (allmap[2:5].permute(1, 2, 0) @ R).permute(2, 0, 1)
    surf = (depth_to_normal(depth[0], intr) @ R).permute(2, 0, 1)
    return {"rgb": rgb, "accum": alpha, "depth": depth, "normal": normal, "surf_normal": surf,
            "dist": allmap[6:7]}


def gaussian_window(channel, dtype):
    g = torch.tensor([math.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)])  # float32
    g = (g / g.sum()).unsqueeze(1)
    return (g @ g.t()).float()[None, None].expand(channel, 1, 11, 11).contiguous().to(dtype)


def ssim_map(x, y):
    c = x.shape[-3]
    w = gaussian_window(c, x.dtype)
    conv = lambda t: F.conv2d(t, w, padding=5, groups=c)
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


def loss_terms(pred, gt):
    """Unweighted (rgb, normal, alpha, depth, dist) terms."""
    sky = gt["rgb"].sum(dim=0) == 0.0
    valid = (~sky) & (gt["depth"].sum(dim=0) > 0.0)
    nv, ns = max(int(valid.sum()), 1), max(int(sky.sum()), 1)
    vf, sf = valid.to(pred["rgb"].dtype), sky.to(pred["rgb"].dtype)
    l1 = ((pred["rgb"] - gt["rgb"]).abs() * vf).sum() / (3 * nv)
    ssim = (ssim_map(pred["rgb"][None], gt["rgb"][None])[0] * vf).sum() / (3 * nv)
    rgb = 0.8 * l1 + 0.2 * (1.0 - ssim)
    normal = ((1 - (pred["normal"] * pred["surf_normal"]).sum(dim=0)) * vf).sum() / nv
    alpha = (pred["accum"] * sf).sum() / ns
    depth = ((pred["depth"] - gt["depth"]).abs() * (1.0 / gt["depth_cov"]) * vf).sum() / nv
    dist = (pred["dist"] * vf).sum() / nv
    return rgb, normal, alpha, depth, dist, int(valid.sum()), int(sky.sum())


def total_loss(pred, gt, weights):
    rgb, normal, alpha, depth, dist, _, _ = loss_terms(pred, gt)
    return (weights["rgb_loss"] * rgb + weights["normal_loss"] * normal +
            weights["alpha_loss"] * alpha + weights["depth_loss"] * depth +
            weights["dist_loss"] * dist)


def loss_and_grads(rgb, allmap, gt, w2c, intr, weights):
    """float64 loss vector [8] (total, rgb, normal, alpha, depth, dist, n_valid, n_sky) and
    d total / d rgb, d total / d allmap.  Pixels whose allmap0 / allmap1 is not finite get
    no gradient through the depth (the reference's graph yields NaN = 0 * inf there; the
    rasteriser's backward never reads those pixels because nothing was blended into them)."""
    rgb = rgb.detach().to("cpu", DT).requires_grad_(True)
    allmap = allmap.detach().to("cpu", DT).requires_grad_(True)
    gt = {k: v.detach().to("cpu", DT) for k, v in gt.items()}
    pred = postprocess(rgb, allmap, w2c.detach().to("cpu", DT), intr)
    terms = loss_terms(pred, gt)
    total = (weights["rgb_loss"] * terms[0] + weights["normal_loss"] * terms[1] +
             weights["alpha_loss"] * terms[2] + weights["depth_loss"] * terms[3] +
             weights["dist_loss"] * terms[4])
    total.backward()
    vec = torch.tensor([float(total.detach())] + [float(torch.as_tensor(t).detach()) for t in terms],
                       dtype=DT)
    return vec, rgb.grad, allmap.grad
