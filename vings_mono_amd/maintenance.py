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


def stablemask_control(mapper):
    """End-of-window stable-mask update and reset of the window scores
    (gaussian_model.py:190-200)."""
    loc = mapper.local_scores
    to_stable = (~mapper.stable_mask) & (loc[:, 0] < 1e-4)
    mapper.stable_mask[to_stable] = True
    to_unstable = (loc[:, 1] > 0.3) & (loc[:, 0] > 0.05) & mapper.stable_mask
    mapper.stable_mask[to_unstable] = False
    mapper.local_scores.mul_(0.0)
