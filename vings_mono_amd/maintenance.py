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
    # multi-GPU jobs: the record maxima not yet merged over the ranks travel with their splats
    pending = getattr(mapper, "_pending_keys", None)
    if pending is not None:
        arrays.append(pending); slots.append(("pending", "keys"))
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
    if pending is not None:
        mapper._pending_keys = new[("pending", "keys")]
    mapper.n = m
    mapper._fb = None          # buffers of the fused view step are sized per splat count
    mapper.drop_view_graphs()  # ... and so is everything a captured view step has baked in
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


def keyframe_pointcloud(c2w, intr, gt_rgb, gt_depth, pred_accum, n_points, accum_thresh,
                        perm=None):
    """Points a new keyframe contributes (get_pointcloud, gaussian_utils.py:19-54).

    gt_rgb [3,H,W], gt_depth [1,H,W], pred_accum [1,H,W] or None (first frame), all on the
    device; c2w 4x4.  The number of samples is n_points scaled by the share of valid-depth
    pixels whose rendered opacity is at most accum_thresh; the samples themselves are drawn
    from ALL valid-depth pixels, as in the reference.  `perm` (a permutation of the valid
    pixels, row-major order) replaces the random draw.  Returns (xyz [n,3], colors [n,3])."""
    L = _lib.lib()
    _lib.require_gpu_tensor(gt_depth, "gt_depth", torch.float32)
    _lib.require_gpu_tensor(gt_rgb, "gt_rgb", torch.float32)
    dev = gt_depth.device
    H, W = gt_depth.shape[-2:]
    depth = gt_depth.reshape(H, W).contiguous()
    rgb = gt_rgb.reshape(3, H, W).contiguous()
    valid = depth > 0
    n_all = int(valid.sum())
    if pred_accum is None:
        n_open = n_all
    else:
        n_open = int((valid & ~(pred_accum.reshape(H, W) > accum_thresh)).sum())
    n = 0 if n_all == 0 else min(int(n_open / n_all * n_points), n_all)
    # the pool is the population n_all counts: depth > 0 (gaussian_utils.py:32,46), which also
    # keeps negative and NaN depths out of the back-projection
    valid_ids = torch.nonzero(valid.reshape(-1)).squeeze(1)            # row-major, int64
    if perm is None:
        perm = torch.randperm(valid_ids.shape[0], device=dev)
    picked = valid_ids[perm.to(dev)[:n]].contiguous()
    xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
    col = torch.empty((n, 3), dtype=torch.float32, device=dev)
    pose = (ctypes.c_float * 16)(*[float(v) for v in c2w.detach().reshape(-1).tolist()])
    _lib.check(L.vings_keyframe_backproject(
        n, H, W, _lib.ptr(picked), _lib.ptr(depth), _lib.ptr(rgb), pose, float(intr["fu"]),
        float(intr["fv"]), float(intr["cu"]), float(intr["cv"]), _lib.ptr(xyz), _lib.ptr(col),
        _lib.stream_ptr(dev)), "vings_keyframe_backproject")
    return xyz, col


def add_new_frame(mapper, frame, accum_thresh, n_points=40000, max_radius=25, perm=None,
                  rotations=None, keep_moments=False):
    """Map update when a keyframe arrives (gaussian_model.py:92-161): drop the splats this
    view draws with a radius above max_radius, render again, zero the opacity map where depth
    or colour disagree with the frame, and add splats back-projected from the frame.

    frame: {'pose' c2w 4x4, 'depth' [H,W,1], 'image' [H,W,3], 'intrinsic' dict}, on the device.
    The reference's other deletion rule (colour error at the splat's pixel) assigns through a
    chained advanced index and therefore changes nothing; it is left out.  After the append
    the reference builds a fresh optimiser (:161), dropping every Adam moment;
    keep_moments=True keeps the old splats' moments instead.  Returns (n_pruned, n_added)."""
    from . import harness
    dev = mapper.device
    intr = frame["intrinsic"]
    c2w = frame["pose"].to(dev).float()
    w2c = torch.linalg.inv(c2w)
    gt_depth = frame["depth"].to(dev).float().permute(2, 0, 1).contiguous()
    gt_rgb = frame["image"].to(dev).float().permute(2, 0, 1).contiguous()
    with torch.no_grad():
        rets = harness.render({k: v.detach() for k, v in mapper.params.items()}, w2c, intr)
        first_rgb = rets["rgb"]
        too_big = rets["radii"] > max_radius
    n_pruned = int(too_big.sum())
    if n_pruned:
        prune(mapper, too_big)
    with torch.no_grad():
        rets = harness.render({k: v.detach() for k, v in mapper.params.items()}, w2c, intr)
        accum = rets["accum"].clone()
        depth_error = (rets["depth"] - gt_depth).abs()
        # the colour error is taken against the render made BEFORE the prune (:131 reads pred_rgb of :102)
        rgb_error = (first_rgb - gt_rgb).abs().sum(dim=0, keepdim=True)
        accum[depth_error > 10 * depth_error.median()] = 0.0
        accum[rgb_error > 0.1] = 0.0
    xyz, col = keyframe_pointcloud(c2w, intr, gt_rgb, gt_depth, accum, n_points, accum_thresh, perm)
    if rotations is None:
        rotations = torch.randn((xyz.shape[0], 4), dtype=torch.float32, device=dev)
    n_added = xyz.shape[0]
    if n_added >= 4:
        add_splats(mapper, new_splat_params(xyz, col, rotations[:n_added], opacity=0.8))
    else:
        n_added = 0
    if not keep_moments:
        mapper.optimizer.state.clear()
    return n_pruned, n_added


def _keyframe_block(K, reducer):
    """(first, count, rows per rank) of the calling rank's contiguous block of K keyframes."""
    if reducer is None or reducer.comm_world() == 1:
        return 0, K, K
    per = -(-K // reducer.comm_world())
    first = min(K, reducer.rank() * per)
    return first, min(K, first + per) - first, per


def _gather_blocks(reducer, maps, first, count, per):
    """All ranks end up with every rank's rows of maps [K, ...] (rank r owns rows r*per ...)."""
    K = maps.shape[0]
    if K == per * reducer.comm_world():
        reducer.all_gather(maps, maps[first:first + per])
        return
    padded = torch.zeros((per * reducer.comm_world(),) + tuple(maps.shape[1:]), dtype=maps.dtype,
                         device=maps.device)
    mine = torch.zeros((per,) + tuple(maps.shape[1:]), dtype=maps.dtype, device=maps.device)
    mine[:count] = maps[first:first + count]
    reducer.all_gather(padded, mine)
    maps.copy_(padded[:K])


def rectify_tracker(mapper, c2w_list, intr, dense_maps, coarse_maps, reducer=None):
    """Tracker-side refresh after a loop closure (loop_rectify.py:415-426): for every corrected
    keyframe pose render the depth and fold it into the tracker's stored maps.
    dense_maps [K,H,W] is updated in place where the rendered depth is positive; coarse_maps
    [K,H/8,W/8] receives the mean of the four centre pixels of every 8x8 block.  Both live on
    the device here (the reference keeps them on the host and copies every render back).

    With a `reducer` of more than one rank (every rank holds the same map and calls this with
    the same arguments) the K renders are dealt out in contiguous blocks and the refreshed maps
    gathered, so every replica of the tracker's buffers ends up with all of them."""
    from . import harness
    L = _lib.lib()
    dev = mapper.device
    _lib.require_gpu_tensor(dense_maps, "dense_maps", torch.float32)
    _lib.require_gpu_tensor(coarse_maps, "coarse_maps", torch.float32)
    assert dense_maps.is_contiguous() and coarse_maps.is_contiguous()
    K, H, W = dense_maps.shape
    assert coarse_maps.shape == (K, H // 8, W // 8) and len(c2w_list) == K
    with torch.no_grad():
        # activations once for all K views; the lean blend pass (depth and alpha are all that is
        # read here), its tiles taken longest list first
        act = synth.activate({k: v.detach() for k, v in mapper.params.items()})
        args = [act[k].contiguous() for k in ("means3D", "colors", "opacities", "scales", "rotations")]
        first, count, per = _keyframe_block(K, reducer)
        for k in range(first, first + count):
            w2c = torch.linalg.inv(c2w_list[k].to(dev).float())
            st, _ = surfel.rasterize_forward(*args, harness.view_settings(w2c, intr), keep_keys=False,
                                             lean=True, order_tiles=True)
            allmap = st.allmap
            _lib.check(L.vings_depth_refresh(H, W, _lib.ptr(allmap[0]), _lib.ptr(allmap[1]),
                                             _lib.ptr(dense_maps[k]), _lib.ptr(coarse_maps[k]),
                                             _lib.stream_ptr(dev)), "vings_depth_refresh")
        if count != K:
            _gather_blocks(reducer, dense_maps, first, count, per)
            _gather_blocks(reducer, coarse_maps, first, count, per)


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
    if getattr(mapper, "_pending_keys", None) is not None:
        mapper._pending_keys = torch.cat((mapper._pending_keys, z(k, 2, dt=torch.int64)))
    mapper.n += k
    mapper._fb = None
    mapper.drop_view_graphs()
    return mapper.n


def calc_score(mapper, c2w_list, kf_ids, intr, reducer=None):
    """Top-3 keyframes of every splat by blend-weight mass (loop_model.py:169-214): returns
    (scores [N,3] descending, kf ids [N,3] int32).  One forward-only importance pass per
    keyframe; a view whose render would stay below 0.1 everywhere is skipped like there.

    With a `reducer` of more than one rank (every rank holds the same map and calls this with
    the same arguments) the keyframes are dealt out in contiguous blocks, every rank scores its
    block, and the gathered records are merged by vings_splat_top3_merge into exactly what the
    pass over all keyframes leaves."""
    from .harness import view_settings
    L = _lib.lib()
    dev, n = mapper.device, mapper.n
    first, count, _ = _keyframe_block(len(c2w_list), reducer)
    sharded = count != len(c2w_list)
    c2w_list, kf_ids = c2w_list[first:first + count], kf_ids[first:first + count]
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
            # the reference's host-side test (loop_model.py:181), taken on the device: the host
            # does not wait for the view's largest colour value
            _lib.check(L.vings_splat_top3_update_gated(
                n, _lib.ptr(cur), int(kf), _lib.ptr(scores3), _lib.ptr(ids3), _lib.ptr(rgb_max), 0.1,
                _lib.stream_ptr(dev)), "vings_splat_top3_update_gated")
        if sharded:
            R = reducer.comm_world()
            all_s = torch.empty((R * n, 3), dtype=torch.float32, device=dev)
            all_i = torch.empty((R * n, 3), dtype=torch.int32, device=dev)
            reducer.all_gather(all_s, scores3)
            reducer.all_gather(all_i, ids3)
            scores3, ids3 = torch.empty_like(scores3), torch.empty_like(ids3)
            _lib.check(L.vings_splat_top3_merge(n, R, _lib.ptr(all_s), _lib.ptr(all_i),
                                                _lib.ptr(scores3), _lib.ptr(ids3),
                                                _lib.stream_ptr(dev)), "vings_splat_top3_merge")
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
    if hasattr(mapper, "note_stable_mask"):
        mapper.note_stable_mask()
