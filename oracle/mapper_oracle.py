"""CPU oracle (numpy) for the per-splat stages of the mapper's view step.

TEST INFRASTRUCTURE ONLY: used by tests/ to check csrc/splat_step.hip; nothing in the product
package imports it.  Restates
  get_property activations   /root/reference/scripts/gaussian/gaussian_base.py:44-48,64-72
                             (sigmoid, exp, torch.nn.functional.normalize)
  weighting_grad             /root/reference/scripts/gaussian/gaussian_utils.py:56-62
  add_records                /root/reference/scripts/gaussian/gaussian_model.py:163-168
  step visibility            /root/reference/scripts/gaussian/gaussian_base.py:385-402
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
