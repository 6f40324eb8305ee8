"""Score-driven map maintenance on the device: importance scoring without a backward pass,
pruning as one stream compaction over every per-splat array, and the stable-mask update.

Counterparts in the reference (all PyTorch around the rasteriser extension there):
  storage_control        /root/reference/scripts/gaussian/gaussian_model.py:228-248
  stablemask_control     /root/reference/scripts/gaussian/gaussian_model.py:187-200
  prune_tensors_from_optimizer / update_records(mode="prune")
                         /root/reference/scripts/gaussian/gaussian_base.py:94-110,
                         /root/reference/scripts/gaussian/gaussian_model.py:178-183
  calc_score_v1 scoring  /root/reference/scripts/loop/loop_model.py:169-214
"""
import ctypes

import torch

from . import _lib, surfel, synth


def compact(arrays, keep):
    """Order-preserving removal of the rows with keep == False from every tensor in `arrays`
    (each [N, ...], same N, on the GPU).  Returns (new tensors, rows kept).  One plan (scan of
    the mask) and one copy launch for all arrays together."""
    L = _lib.lib()
    _lib.require_gpu_tensor(keep, "keep")
    n = int(keep.numel())
    dev = keep.device
    keep8 = keep.reshape(-1).to(torch.uint8).contiguous()
    for a in arrays:
        _lib.require_gpu_tensor(a, "array")
        if a.shape[0] != n:
            raise _lib.VingsError("every array must have one row per mask entry")
    if n == 0:
        return [a.clone() for a in arrays], 0
    if len(arrays) > 24:
        raise _lib.VingsError("at most 24 arrays per compaction")
    dst_index = torch.empty(n, dtype=torch.int32, device=dev)
    ws = torch.empty(int(L.vings_splat_compact_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    n_keep = ctypes.c_int64(0)
    stream = _lib.stream_ptr(dev)
    _lib.check(L.vings_splat_compact_plan(n, _lib.ptr(keep8), _lib.ptr(dst_index), _lib.ptr(ws),
                                          ws.numel(), ctypes.byref(n_keep), stream),
               "vings_splat_compact_plan")
    m = int(n_keep.value)
    srcs = [a.contiguous() for a in arrays]
    outs = [torch.empty((m,) + tuple(a.shape[1:]), dtype=a.dtype, device=dev) for a in srcs]
    if m > 0 and srcs:
        k = len(srcs)
        row_bytes = [a.element_size() * (a.numel() // n) for a in srcs]
        S = (ctypes.c_void_p * k)(*[_lib.ptr(a) for a in srcs])
        D = (ctypes.c_void_p * k)(*[_lib.ptr(o) for o in outs])
        R = (ctypes.c_int * k)(*row_bytes)
        _lib.check(L.vings_splat_compact_apply(n, _lib.ptr(dst_index), k, S, D, R, stream),
                   "vings_splat_compact_apply")
    return outs, m


def prune(mapper, prune_mask):
    """Remove the splats with prune_mask == True from a harness.SurfelMapper: parameters,
    Adam moments (kept under the names exp_avg / exp_avg_sq like the reference edits them)
    and the score records.  Returns the number of splats left."""
    names = mapper.NAMES
    arrays, slots = [], []
    for k in names:
        p = mapper.params[k]
        arrays.append(p.detach()); slots.append(("param", k))
        st = mapper.optimizer.state.get(p, None)
        if st:
            arrays.append(st["exp_avg"]); slots.append(("exp_avg", k))
            arrays.append(st["exp_avg_sq"]); slots.append(("exp_avg_sq", k))
    for attr in ("global_scores", "local_scores", "stable_mask", "globalkf_id", "globalkf_max_scores"):
        arrays.append(getattr(mapper, attr)); slots.append(("record", attr))
    outs, m = compact(arrays, ~prune_mask.reshape(-1).bool())
    new = dict(zip(slots, outs))
    for group in mapper.optimizer.param_groups:
        k = group["name"]
        old = group["params"][0]
        state = mapper.optimizer.state.pop(old, None)
        p = torch.nn.Parameter(new[("param", k)].requires_grad_(True))
        group["params"][0] = p
        mapper.params[k] = p
        if state:
            state["exp_avg"] = new[("exp_avg", k)]
            state["exp_avg_sq"] = new[("exp_avg_sq", k)]
            mapper.optimizer.state[p] = state
    for attr in ("global_scores", "local_scores", "stable_mask", "globalkf_id", "globalkf_max_scores"):
        setattr(mapper, attr, new[("record", attr)])
    mapper.n = m
    mapper._fb = None          # buffers of the fused view step are sized per splat count
    return m


def importance_over_views(mapper, w2c_list, intr, pixel_mask=None):
    """Sum over the given keyframes of every splat's blend-weight mass (forward passes only)."""
    from .harness import view_settings
    with torch.no_grad():
        act = synth.activate({k: v.detach() for k, v in mapper.params.items()})
        args = [act[k].contiguous() for k in ("means3D", "colors", "opacities", "scales", "rotations")]
        total = torch.zeros(mapper.n, dtype=torch.float32, device=mapper.device)
        for w2c in w2c_list:
            surfel.view_importance(*args, view_settings(w2c, intr, pixel_mask=pixel_mask), out=total,
                                   accumulate=True)
    return total


def storage_control(mapper, w2c_list, intr, low=0.05, high=0.8):
    """Re-score every splat over the keyframe list and prune the unstable ones whose summed
    importance lies in (low, high) (gaussian_model.py:233-248).  Returns the prune mask."""
    scores = importance_over_views(mapper, w2c_list, intr)
    mask = (scores > low) & (~mapper.stable_mask) & (scores < high)
    prune(mapper, mask)
    return mask


def mean_dist2_to_3nn(points):
    """distCUDA2 (gaussian_utils.py:12-16) on the device: mean squared distance of every
    point [N,3] to its three nearest other points."""
    L = _lib.lib()
    _lib.require_gpu_tensor(points, "points", torch.float32)
    pts = points.contiguous()
    out = torch.empty(pts.shape[0], dtype=torch.float32, device=pts.device)
    _lib.check(L.vings_knn3_mean_dist2(pts.shape[0], _lib.ptr(pts), _lib.ptr(out),
                                       _lib.stream_ptr(pts.device)), "vings_knn3_mean_dist2")
    return out


def new_splat_params(points, colors, rotations, opacity=0.8):
    """Stored (pre-activation) parameters of splats created from a point cloud, as
    add_new_frame / init_first_frame set them (gaussian_model.py:72-80,139-149): isotropic
    log-scale from the 3-nearest-neighbour distance, opacity logit of `opacity`."""
    dist2 = mean_dist2_to_3nn(points).clamp_min(0.0000001)
    log_scales = torch.log(torch.sqrt(dist2))[:, None].repeat(1, 2)
    logit = float(torch.log(torch.tensor(opacity / (1.0 - opacity))))
    n = points.shape[0]
    return {"_xyz": points.float().contiguous(), "_rgb": colors.float().contiguous(),
            "_scaling": log_scales.contiguous(), "_rotation": rotations.float().contiguous(),
            "_opacity": torch.full((n, 1), logit, dtype=torch.float32, device=points.device)}


def add_splats(mapper, new_params):
    """Append splats to a harness.SurfelMapper: parameters concatenated, Adam moments and
    score records extended with zeros (cat_tensors_to_optimizer gaussian_base.py:74-92,
    update_records(mode="densify") gaussian_model.py:171-176)."""
    k = new_params["_xyz"].shape[0]
    dev = mapper.device
    for group in mapper.optimizer.param_groups:
        name = group["name"]
        old = group["params"][0]
        ext = new_params[name].to(dev).float().reshape((k,) + tuple(old.shape[1:]))
        state = mapper.optimizer.state.pop(old, None)
        p = torch.nn.Parameter(torch.cat((old.detach(), ext), dim=0).requires_grad_(True))
        group["params"][0] = p
        mapper.params[name] = p
        if state:
            state["exp_avg"] = torch.cat((state["exp_avg"], torch.zeros_like(ext)), dim=0)
            state["exp_avg_sq"] = torch.cat((state["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            mapper.optimizer.state[p] = state
    z = lambda *s, dt: torch.zeros(*s, dtype=dt, device=dev)
    mapper.global_scores = torch.cat((mapper.global_scores, z(k, 2, dt=torch.float32)))
    mapper.local_scores = torch.cat((mapper.local_scores, z(k, 2, dt=torch.float32)))
    mapper.stable_mask = torch.cat((mapper.stable_mask, z(k, dt=torch.bool)))
    mapper.globalkf_id = torch.cat((mapper.globalkf_id, z(k, dt=torch.int64)))
    mapper.globalkf_max_scores = torch.cat((mapper.globalkf_max_scores, z(k, dt=torch.float32)))
    mapper.n += k
    mapper._fb = None
    return mapper.n


def calc_score(mapper, c2w_list, kf_ids, intr):
    """Top-3 keyframes of every splat by blend-weight mass (loop_model.py:169-214): returns
    (scores [N,3] descending, kf ids [N,3] int32).  One forward-only importance pass per
    keyframe; a view whose render would stay below 0.1 everywhere is skipped like there."""
    from .harness import view_settings
    L = _lib.lib()
    dev, n = mapper.device, mapper.n
    scores3 = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    ids3 = torch.zeros((n, 3), dtype=torch.int32, device=dev)
    cur = torch.empty(n, dtype=torch.float32, device=dev)
    rgb_max = torch.zeros(1, dtype=torch.float32, device=dev)
    with torch.no_grad():
        act = synth.activate({k: v.detach() for k, v in mapper.params.items()})
        args = [act[k].contiguous() for k in ("means3D", "colors", "opacities", "scales", "rotations")]
        for c2w, kf in zip(c2w_list, kf_ids):
            w2c = torch.linalg.inv(c2w.to(dev))
            surfel.view_importance(*args, view_settings(w2c, intr), out=cur, accumulate=False,
                                   rgb_max=rgb_max)
            if float(rgb_max) < 0.1:      # the reference's own host-side test (loop_model.py:181)
                continue
            _lib.check(L.vings_splat_top3_update(n, _lib.ptr(cur), int(kf), _lib.ptr(scores3),
                                                 _lib.ptr(ids3), _lib.stream_ptr(dev)),
                       "vings_splat_top3_update")
    return scores3, ids3


def _quat_wxyz_from_matrix(R):
    """Unit quaternions (w,x,y,z) of rotation matrices [K,3,3] (float64 on the host)."""
    R = R.double().cpu()
    K = R.shape[0]
    q = torch.zeros(K, 4, dtype=torch.float64)
    for k in range(K):
        m = R[k]
        t = m[0, 0] + m[1, 1] + m[2, 2]
        if t > 0:
            s = torch.sqrt(t + 1.0) * 2
            q[k] = torch.stack([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s,
                                (m[1, 0] - m[0, 1]) / s])
        elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
            s = torch.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
            q[k] = torch.stack([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s,
                                (m[0, 2] + m[2, 0]) / s])
        elif m[1, 1] > m[2, 2]:
            s = torch.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
            q[k] = torch.stack([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s,
                                (m[1, 2] + m[2, 1]) / s])
        else:
            s = torch.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
            q[k] = torch.stack([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s,
                                (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    return q / q.norm(dim=1, keepdim=True)


def rectify_gaussians(mapper, raw_c2ws, new_c2ws, new_scales, start_kf, end_kf, scores3, ids3,
                      delete_below=0.1):
    """Move the splats owned (best keyframe) by keyframes start_kf..end_kf with their
    keyframe's pose correction and per-keyframe scale, then delete the splats whose best score
    is below `delete_below` (loop_rectify.py:300-382).  raw_c2ws / new_c2ws: [K,4,4];
    new_scales: one scale per keyframe of the loop (end_kf - start_kf + 1 values).
    Rotations are handled as (w,x,y,z) quaternions (the reference's round trip through
    (x,y,z,w) libraries is not reproduced)."""
    L = _lib.lib()
    dev, n = mapper.device, mapper.n
    raw = raw_c2ws.double().cpu()
    new = new_c2ws.double().cpu()
    K = raw.shape[0]
    old_w2c = torch.linalg.inv(raw)
    delta = new @ old_w2c
    dq = _quat_wxyz_from_matrix(delta[:, :3, :3])
    scale = torch.ones(K, dtype=torch.float64)
    scale[start_kf:end_kf + 1] = new_scales.double().cpu().reshape(-1)
    f = lambda t: t.float().contiguous().to(dev)
    t_old, t_new = f(old_w2c[:, :3, :].reshape(K, 12)), f(new[:, :3, :].reshape(K, 12))
    t_dq, t_scale = f(dq), f(scale)
    ids3 = ids3.contiguous()
    with torch.no_grad():
        _lib.check(L.vings_splat_reanchor(
            n, _lib.ptr(mapper.params["_xyz"].data), _lib.ptr(mapper.params["_rotation"].data),
            _lib.ptr(ids3), 3, int(start_kf), int(end_kf), K, _lib.ptr(t_old), _lib.ptr(t_new),
            _lib.ptr(t_dq), _lib.ptr(t_scale), _lib.stream_ptr(dev)), "vings_splat_reanchor")
    delete_mask = scores3[:, 0] < delete_below
    prune(mapper, delete_mask)
    return delete_mask


def stablemask_control(mapper):
    """End-of-window stable-mask update and reset of the window scores
    (gaussian_model.py:190-200)."""
    loc = mapper.local_scores
    to_stable = (~mapper.stable_mask) & (loc[:, 0] < 1e-4)
    mapper.stable_mask[to_stable] = True
    to_unstable = (loc[:, 1] > 0.3) & (loc[:, 0] > 0.05) & mapper.stable_mask
    mapper.stable_mask[to_unstable] = False
    mapper.local_scores.mul_(0.0)
