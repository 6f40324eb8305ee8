"""CPU oracle (numpy) for the per-splat stages of the mapper's view step.

TEST INFRASTRUCTURE ONLY: used by tests/ to check csrc/splat_step.hip; nothing in the product
package imports it.  Restates
  get_property activations   /root/reference/scripts/gaussian/gaussian_base.py:44-48,64-72
                             (sigmoid, exp, torch.nn.functional.normalize)
  weighting_grad             /root/reference/scripts/gaussian/gaussian_utils.py:56-62
  add_records                /root/reference/scripts/gaussian/gaussian_model.py:163-168
  step visibility            /root/reference/scripts/gaussian/gaussian_base.py:385-402
  calc_score_v1 top-3 record /root/reference/scripts/loop/loop_model.py:190-211
  rectify_gaussians          /root/reference/scripts/loop/loop_rectify.py:322-367 (with
                             (w,x,y,z) quaternions; parity unpinned: lietorch is absent)
`weight` is pinned by the golden vectors captured from the reference's weighting_grad
(tests/golden/harness_golden.npz, checked in tests/test_oracle_loss.py); the activations and
their derivatives are the textbook ones and are checked by finite differences there.
"""
import numpy as np


def activate(raw_opacity, raw_scaling, raw_rotation):
    opac = 1.0 / (1.0 + np.exp(-raw_opacity))
    scales = np.exp(raw_scaling)
    nrm = np.maximum(np.linalg.norm(raw_rotation, axis=1, keepdims=True), 1e-12)
    return opac, scales, raw_rotation / nrm


def weight(cur_scores, global_scores):
    """Per-splat gradient weight with the global score already holding this view:
    s0 / (S0 + 1e-6 + s0)."""
    s0 = cur_scores[:, 0]
    return s0 / (global_scores[:, 0] + 1e-6 + s0)


def top3_update(scores3, ids3, cur, kf):
    """One keyframe of calc_score_v1 (/root/reference/scripts/loop/loop_model.py:190-211):
    masks from the record as it was, then the three shifts applied in order."""
    s, k = scores3.copy(), ids3.copy()
    first = s[:, 0] < cur
    second = (s[:, 0] > cur) & (s[:, 1] < cur)
    third = (s[:, 1] > cur) & (s[:, 2] < cur)
    s[first, 2] = s[first, 1]; s[first, 1] = s[first, 0]; s[first, 0] = cur[first]
    k[first, 2] = k[first, 1]; k[first, 1] = k[first, 0]; k[first, 0] = kf
    s[second, 2] = s[second, 1]; s[second, 1] = cur[second]
    k[second, 2] = k[second, 1]; k[second, 1] = kf
    s[third, 2] = cur[third]; k[third, 2] = kf
    return s, k


def quat_to_matrix_wxyz(q):
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], 1),
        np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], 1),
        np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1)], 1)


def reanchor(xyz, rot_wxyz, owner, start, end, raw_c2ws, new_c2ws, scales):
    """rectify_gaussians (/root/reference/scripts/loop/loop_rectify.py:322-367) in float64:
    returns (new xyz, new rotation MATRICES [N,3,3]) -- matrices so that the check does not
    depend on a quaternion sign or extraction branch."""
    xyz = xyz.astype(np.float64).copy()
    R = quat_to_matrix_wxyz(rot_wxyz.astype(np.float64))
    upd = (owner >= start) & (owner <= end)
    k = owner[upd]
    old_w2c = np.linalg.inv(raw_c2ws.astype(np.float64))
    T = new_c2ws.astype(np.float64) @ old_w2c
    cam = np.einsum("nij,nj->ni", old_w2c[k, :3, :3], xyz[upd]) + old_w2c[k, :3, 3]
    cam = cam * scales.astype(np.float64)[k - start].reshape(-1, 1)
    xyz[upd] = np.einsum("nij,nj->ni", new_c2ws[k, :3, :3].astype(np.float64), cam) + new_c2ws[k, :3, 3]
    R[upd] = T[k, :3, :3] @ R[upd]
    return xyz, R


def grad_finish(opac, scales, raw_rotation, d, cur, radii, stable, glob, loc, kf_max, kf_id,
                this_kf):
    """d: dict of gradients w.r.t. activated values (means3D, colors, opacities, scales,
    rotations).  Returns (grads w.r.t. stored parameters, visible, updated score arrays)."""
    dt = opac.dtype
    s0, s1 = cur[:, 0], cur[:, 1]
    g0_new = np.clip(glob[:, 0] + s0, 0.0, 1e4).astype(dt)
    glob2, loc2 = glob.copy(), loc.copy()
    glob2[:, 0] = g0_new
    w = weight(cur, glob2).astype(dt)[:, None]
    loc2[:, 0] += s0
    loc2[:, 1] = np.maximum(loc2[:, 1], s1)
    replace = kf_max < s0
    kf_max2 = np.where(replace, s0, kf_max)
    kf_id2 = np.where(replace, this_kf, kf_id)
    q = raw_rotation
    ln = np.linalg.norm(q, axis=1, keepdims=True)
    n = q / np.maximum(ln, 1e-12)
    g = d["rotations"]
    g_rot = (g - n * np.sum(n * g, axis=1, keepdims=True)) / np.maximum(ln, 1e-12)
    out = {
        "_xyz": d["means3D"] * w,
        "_rgb": d["colors"] * w,
        "_opacity": d["opacities"].reshape(-1, 1) * (opac * (1 - opac)).reshape(-1, 1) * w,
        "_scaling": d["scales"] * scales * w,
        "_rotation": g_rot * w,
    }
    visible = (radii > 0) & ~stable
    return out, visible, glob2, loc2, kf_max2, kf_id2
