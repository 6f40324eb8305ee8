"""Host-side mirror of the reference's ``diff_surfel_rasterization`` extension.

Same names, argument meaning and error behaviour as the module the reference
imports at /root/reference/scripts/gaussian/gaussian_base.py:5-6:
``GaussianRasterizationSettings``, ``GaussianRasterizer`` and
``SparseGaussianAdam``.  All compute happens in libvings_hip.so through the C
ABI of include/vings_hip.h; torch only provides device memory, the current
stream and the autograd hook.
"""
import ctypes
import os
from typing import NamedTuple, Optional

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

GEOM_STRIDE = 24
GRAD_STRIDE = 20
LAST_INSTANCE_COUNT = 0  # D of the most recent forward pass (diagnostics)
AUX_CHANNELS = 9
TILE = 16


class GaussianRasterizationSettings(NamedTuple):
    """Fields as constructed at gaussian_base.py:160-177 (pixel_mask optional,
    sky_utils.py:135-149; u2_minus_u1 optional, gaussian_model.py:268-286)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    pixel_mask: Optional[torch.Tensor] = None
    u2_minus_u1: Optional[torch.Tensor] = None


def _f32c(t, name):
    _lib.require_gpu_tensor(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class SurfelForwardState:
    """Device buffers produced by one forward pass (kept for backward / tests)."""
    __slots__ = ("N", "H", "W", "D", "geom", "depth", "radii", "rect", "tiles_touched",
                 "block_prefix", "keys", "vals", "ranges", "tile_order", "final_T", "n_contrib", "color",
                 "allmap", "tail_ptr", "_keep")

    def __init__(self):
        self.tile_order = None      # set when the blend passes take their tiles in a chosen order


def set_tight_binning(on):
    """Bin splats only into the tiles their alpha >= 1/255 box reaches (True, the default) or
    into the reference's full 3-sigma square (False: instance list identical to the
    reference's).  Rendered values and gradients are the same either way.  Returns the
    previous setting."""
    return bool(_lib.lib().vings_surfel_set_tight_binning(1 if on else 0))


class thread_binning:
    """`with thread_binning(tight=False): ...` — the binning mode of the projections the
    calling thread launches inside the block, whatever the process-wide switch says and
    without touching it (other threads rendering on the same GPU keep their own mode)."""

    def __init__(self, tight):
        self.mode = 1 if tight else 0

    def __enter__(self):
        self.prev = _lib.lib().vings_surfel_set_tight_binning_thread(self.mode)
        return self

    def __exit__(self, *exc):
        _lib.lib().vings_surfel_set_tight_binning_thread(self.prev)
        return False


def rasterize_forward(means3D, colors, opacities, scales, rotations, settings, keep_keys=False,
                      raw=None, dgeom=None, lean=False, order_tiles=None):
    """Run projection -> binning -> blend. Returns (SurfelForwardState, aux).

    `raw` = (raw_opacity, raw_scaling, raw_rotation): the stored parameters.  The projection
    then applies the mapper's activations itself and FILLS opacities / scales / rotations
    (which only need to be allocated) instead of reading them.  With `raw`, `dgeom`
    ([N, GRAD_STRIDE] float32) may be handed over too: the projection zeroes the rows the
    backward pass will add into, and `blend_backward(..., dgeom=dgeom)` skips its own clearing.
    `lean`: the blend pass of the mapper's own view step, which leaves the median depth and the
    flow planes at zero (vings_surfel_blend_forward_lean).
    `order_tiles`: the blend pass takes its tiles longest list first (default: whenever `dgeom`
    is given, i.e. in the mapper's view step)."""
    if lean and settings.u2_minus_u1 is not None:
        raise _lib.VingsError("the lean blend passes carry no flow")
    return _forward(means3D, colors, opacities, scales, rotations, settings, keep_keys, True, raw,
                    dgeom, lean, order_tiles)


def view_importance(means3D, colors, opacities, scales, rotations, settings, out=None,
                    accumulate=False, rgb_max=None):
    """Per-splat blend weight mass sum(alpha*T) of one view without rendering an image or
    running a backward pass (= scores.grad[:,0] of the reference's render + backward,
    gaussian_model.py:233-241).  Returns (importance[N], radii[N]).  `rgb_max`: optional
    1-element float tensor that receives the largest colour value the view would render."""
    L = _lib.lib()
    st, aux = _forward(means3D, colors, opacities, scales, rotations, settings, False, False)
    N = st.N
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=means3D.device)
        accumulate = False
    mask = aux[3]
    order = None
    if ORDER_TILES and N > 0:
        order = torch.empty(st.ranges.shape[0], dtype=torch.int32, device=means3D.device)
        check(L.vings_surfel_tile_order(st.H, st.W, ptr(st.ranges), ptr(order),
                                        stream_ptr(means3D.device)), "vings_surfel_tile_order")
    check(L.vings_surfel_blend_importance_ordered(
        N, st.H, st.W, ptr(st.ranges), ptr(st.vals), ptr(st.geom), ptr(mask), ptr(out),
        1 if accumulate else 0, ptr(aux[2]), ptr(rgb_max), ptr(order), stream_ptr(means3D.device)),
        "vings_surfel_blend_importance_ordered")
    return out, st.radii


ORDER_TILES = os.environ.get("VINGS_ORDER_TILES", "1") != "0"   # blend passes of the view step take their tiles longest list first
SPECULATE = os.environ.get("VINGS_SPECULATE", "1") != "0"   # launch sort + blend before the host has the instance count (see _forward)
_D_HINT = {}       # (device index, H, W) -> (instance count, depth key bits) of the latest view


def _forward(means3D, colors, opacities, scales, rotations, settings, keep_keys, blend, raw=None,
             dgeom=None, lean=False, order_tiles=None):
    L = _lib.lib()
    dev = means3D.device
    N = means3D.shape[0]
    H, W = int(settings.image_height), int(settings.image_width)
    if dgeom is not None and raw is None:
        raise _lib.VingsError("dgeom is cleared by the projection only together with raw=...")
    st = SurfelForwardState()
    st.N, st.H, st.W = N, H, W
    view = _f32c(settings.viewmatrix, "viewmatrix")
    proj = _f32c(settings.projmatrix, "projmatrix")
    bg = _f32c(settings.bg, "bg")
    flow = settings.u2_minus_u1
    if flow is not None:
        flow = _f32c(flow, "u2_minus_u1")
        if flow.shape[0] != N:
            raise _lib.VingsError("u2_minus_u1 must be [N,2]")
    mask = settings.pixel_mask
    if mask is not None:
        _lib.require_gpu_tensor(mask, "pixel_mask")
        if mask.numel() != H * W:
            raise _lib.VingsError("pixel_mask must hold H*W entries")
        mask = mask.reshape(-1).to(torch.uint8).contiguous()
    stream = stream_ptr(dev)
    nb = (N + 255) // 256
    st.geom = torch.empty((max(N, 1), GEOM_STRIDE), dtype=torch.float32, device=dev)
    st.depth = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
    st.radii = torch.empty(N, dtype=torch.int32, device=dev)   # written for every splat
    st.rect = torch.empty((max(N, 1), 4), dtype=torch.int32, device=dev)
    st.tiles_touched = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
    st.block_prefix = torch.empty(3 * nb + 4, dtype=torch.int32, device=dev)
    T = ((W + TILE - 1) // TILE) * ((H + TILE - 1) // TILE)
    st.ranges = torch.empty((T, 2), dtype=torch.int32, device=dev)
    aux = (view, proj, bg, mask, flow)
    project_args = (N, H, W, float(settings.scale_modifier), ptr(means3D), ptr(scales),
                    ptr(rotations), ptr(opacities), ptr(colors), ptr(flow), ptr(view), ptr(proj),
                    ptr(st.geom), ptr(st.depth), ptr(st.radii), ptr(st.rect),
                    ptr(st.tiles_touched), ptr(st.block_prefix))
    d_host = ctypes.c_int64(0)
    d_range = (ctypes.c_uint32 * 2)(0, 0xFFFFFFFF)

    def launch_projection():
        if raw is None:
            check(L.vings_surfel_project_launch(*project_args, stream), "vings_surfel_project_launch")
        else:
            raw_o, raw_s, raw_r = raw
            check(L.vings_surfel_project_launch_activate(
                N, H, W, float(settings.scale_modifier), ptr(means3D), ptr(raw_s), ptr(raw_r),
                ptr(raw_o), ptr(colors), ptr(flow), ptr(view), ptr(proj), ptr(scales),
                ptr(rotations), ptr(opacities), *project_args[12:], ptr(dgeom), stream),
                "vings_surfel_project_launch_activate")

    def blend_forward():
        st.color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        st.allmap = torch.empty((AUX_CHANNELS, H, W), dtype=torch.float32, device=dev)
        st.final_T = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        st.n_contrib = torch.empty((2, H, W), dtype=torch.int32, device=dev)
        if ORDER_TILES and (dgeom is not None if order_tiles is None else order_tiles):
            # the mapper's view step: both blend passes take the tiles longest list first
            st.tile_order = torch.empty(T, dtype=torch.int32, device=dev)
            check(L.vings_surfel_tile_order(H, W, ptr(st.ranges), ptr(st.tile_order), stream),
                  "vings_surfel_tile_order")
            check(L.vings_surfel_blend_forward_ordered(
                1 if lean else 0, H, W, ptr(st.ranges), ptr(st.vals), ptr(st.geom), ptr(bg),
                ptr(mask), ptr(st.color), ptr(st.allmap), ptr(st.final_T), ptr(st.n_contrib),
                ptr(st.tile_order), stream), "vings_surfel_blend_forward_ordered")
            return
        fwd = L.vings_surfel_blend_forward_lean if lean else L.vings_surfel_blend_forward
        check(fwd(H, W, ptr(st.ranges), ptr(st.vals), ptr(st.geom), ptr(bg), ptr(mask),
                  ptr(st.color), ptr(st.allmap), ptr(st.final_T), ptr(st.n_contrib), stream),
              "vings_surfel_blend_forward")

    # The instance count D comes back from the projection through the host.  Waiting for it
    # before the sort is launched leaves the GPU idle once per view, so when an earlier view of
    # this size is known the sort and the blend are launched at once, sized from that view's
    # count plus 1/8; the kernels read the true count from device memory.  The host then checks
    # the guess and, if it was too small, repeats the two stages with exact sizes (nothing but
    # their own outputs was written).
    hint = _D_HINT.get((dev.index, H, W)) if SPECULATE and N > 0 else None
    done = False
    if hint is not None and hint[0] > 0:
        cap = hint[0] + hint[0] // 8 + 4096
        ws_bytes = int(L.vings_surfel_bin_workspace_bytes(cap))
        st.keys = torch.empty(cap, dtype=torch.int64, device=dev)
        st.vals = torch.empty(cap, dtype=torch.int32, device=dev)
        ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
        launch_projection()
        tail = st.block_prefix.data_ptr() + 4 * (3 * nb + 1)
        check(L.vings_surfel_bin_speculative(
            N, H, W, cap, hint[1], tail, ptr(st.depth), ptr(st.rect), ptr(st.tiles_touched),
            ptr(st.block_prefix), ptr(st.keys), ptr(st.vals), ptr(st.ranges),
            (ws.data_ptr() + 255) & ~255, ws_bytes, stream), "vings_surfel_bin_speculative")
        if blend:
            blend_forward()
        check(L.vings_surfel_project_wait(ctypes.byref(d_host), d_range), "vings_surfel_project_wait")
        D = int(d_host.value)
        bits = int(L.vings_surfel_bin_low_bits(int(d_range[0]), int(d_range[1]))) if D > 0 else 0
        done = D <= cap
        _D_HINT[(dev.index, H, W)] = (D, max(bits, hint[1])) if D > 0 else hint
    else:
        if raw is None:
            check(L.vings_surfel_project(*project_args, ctypes.byref(d_host), d_range, stream),
                  "vings_surfel_project")
        elif N > 0:
            launch_projection()
            check(L.vings_surfel_project_wait(ctypes.byref(d_host), d_range), "vings_surfel_project_wait")
        D = int(d_host.value)
        if D > 0:
            _D_HINT[(dev.index, H, W)] = (
                D, int(L.vings_surfel_bin_low_bits(int(d_range[0]), int(d_range[1]))))
    st.D = D
    global LAST_INSTANCE_COUNT
    LAST_INSTANCE_COUNT = D
    if not done:
        st.keys = torch.empty(max(D, 1), dtype=torch.int64, device=dev)
        st.vals = torch.empty(max(D, 1), dtype=torch.int32, device=dev)
        ws_bytes = int(L.vings_surfel_bin_workspace_bytes(D))
        ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
        check(L.vings_surfel_bin(N, H, W, D, int(d_range[0]), int(d_range[1]), ptr(st.depth),
                                 ptr(st.rect), ptr(st.tiles_touched), ptr(st.block_prefix),
                                 ptr(st.keys), ptr(st.vals), ptr(st.ranges),
                                 (ws.data_ptr() + 255) & ~255, ws_bytes, stream), "vings_surfel_bin")
        if blend:
            blend_forward()
    if not blend or not keep_keys:
        st.keys = None
    return st, aux


def rasterize_forward_fixed(means3D, colors, opacities, scales, rotations, settings, raw, dgeom,
                            capacity, count_words):
    """The forward pass of `rasterize_forward(raw=..., dgeom=...)` with nothing in it that a
    stream capture cannot record: the instance list is binned into buffers of `capacity`
    entries and the host does not wait for the count.  The projection leaves {depth bits min,
    max, instance count D} in `count_words` (uint32 [>=3], pinned host memory or device memory);
    the result is only valid if D <= capacity, which the CALLER checks later (and the guarded
    finish / Adam entry points check on the device)."""
    L = _lib.lib()
    dev = means3D.device
    N = means3D.shape[0]
    H, W = int(settings.image_height), int(settings.image_width)
    if N == 0 or capacity <= 0:
        raise _lib.VingsError("rasterize_forward_fixed needs splats and a positive capacity")
    st = SurfelForwardState()
    st.N, st.H, st.W = N, H, W
    view = _f32c(settings.viewmatrix, "viewmatrix")
    proj = _f32c(settings.projmatrix, "projmatrix")
    bg = _f32c(settings.bg, "bg")
    flow = settings.u2_minus_u1
    if flow is not None:
        flow = _f32c(flow, "u2_minus_u1")
    mask = settings.pixel_mask
    if mask is not None:
        mask = mask.reshape(-1).to(torch.uint8).contiguous()
    stream = stream_ptr(dev)
    nb = (N + 255) // 256
    st.geom = torch.empty((N, GEOM_STRIDE), dtype=torch.float32, device=dev)
    st.depth = torch.empty(N, dtype=torch.float32, device=dev)
    st.radii = torch.empty(N, dtype=torch.int32, device=dev)
    st.rect = torch.empty((N, 4), dtype=torch.int32, device=dev)
    st.tiles_touched = torch.empty(N, dtype=torch.int32, device=dev)
    st.block_prefix = torch.empty(3 * nb + 4, dtype=torch.int32, device=dev)
    T = ((W + TILE - 1) // TILE) * ((H + TILE - 1) // TILE)
    st.ranges = torch.empty((T, 2), dtype=torch.int32, device=dev)
    raw_o, raw_s, raw_r = raw
    check(L.vings_surfel_project_enqueue_activate(
        N, H, W, float(settings.scale_modifier), ptr(means3D), ptr(raw_s), ptr(raw_r), ptr(raw_o),
        ptr(colors), ptr(flow), ptr(view), ptr(proj), ptr(scales), ptr(rotations), ptr(opacities),
        ptr(st.geom), ptr(st.depth), ptr(st.radii), ptr(st.rect), ptr(st.tiles_touched),
        ptr(st.block_prefix), ptr(dgeom), count_words.data_ptr(), stream),
        "vings_surfel_project_enqueue_activate")
    ws_bytes = int(L.vings_surfel_bin_workspace_bytes(capacity))
    st.keys = torch.empty(capacity, dtype=torch.int64, device=dev)
    st.vals = torch.empty(capacity, dtype=torch.int32, device=dev)
    ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
    tail = st.block_prefix.data_ptr() + 4 * (3 * nb + 1)
    check(L.vings_surfel_bin_speculative(
        N, H, W, capacity, 32, tail, ptr(st.depth), ptr(st.rect), ptr(st.tiles_touched),
        ptr(st.block_prefix), ptr(st.keys), ptr(st.vals), ptr(st.ranges),
        (ws.data_ptr() + 255) & ~255, ws_bytes, stream), "vings_surfel_bin_speculative")
    st.color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    st.allmap = torch.empty((AUX_CHANNELS, H, W), dtype=torch.float32, device=dev)
    st.final_T = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    st.n_contrib = torch.empty((2, H, W), dtype=torch.int32, device=dev)
    if ORDER_TILES:
        st.tile_order = torch.empty(T, dtype=torch.int32, device=dev)
        check(L.vings_surfel_tile_order(H, W, ptr(st.ranges), ptr(st.tile_order), stream),
              "vings_surfel_tile_order")
    check(L.vings_surfel_blend_forward_ordered(
        0 if flow is not None else 1, H, W, ptr(st.ranges), ptr(st.vals), ptr(st.geom), ptr(bg),
        ptr(mask), ptr(st.color), ptr(st.allmap), ptr(st.final_T), ptr(st.n_contrib),
        ptr(st.tile_order), stream), "vings_surfel_blend_forward_ordered")
    st.D = None          # on the device / in count_words only
    st.keys = None
    st.tail_ptr = tail   # device address of the count words (for the guarded entry points)
    st._keep = (ws,)
    return st, (view, proj, bg, mask, flow)


def blend_backward(st, aux, g_color, g_allmap, dgeom=None, lean=False, stable=None):
    """Pixel half of the backward pass: the per-splat accumulator rows dgeom [N, GRAD_STRIDE]
    that vings_surfel_project_backward (or its fused _finish form) turns into gradients.
    `dgeom`: rows already cleared by the forward pass (rasterize_forward(dgeom=...)); the pass
    then adds into them instead of clearing a fresh array.
    `stable` ([N] uint8 / bool, needs dgeom): splats the optimiser will not step; they keep
    their place in the compositing and their two scores, their gradient components stay zero
    (vings_surfel_blend_backward_accumulate_stable)."""
    L = _lib.lib()
    _, _, bg, mask, _ = aux
    N, H, W = st.N, st.H, st.W
    dev = st.geom.device
    if lean and dgeom is None:
        raise _lib.VingsError("the lean backward pass adds into rows the forward pass cleared: pass dgeom")
    if stable is not None:
        if dgeom is None:
            raise _lib.VingsError("the stable-splat backward pass adds into cleared rows: pass dgeom")
        _lib.require_gpu_tensor(stable, "stable")
        if stable.numel() != N or stable.element_size() != 1:
            raise _lib.VingsError("stable must hold N one-byte flags")
        check(L.vings_surfel_blend_backward_accumulate_stable(
            1 if lean else 0, N, H, W, ptr(st.ranges), ptr(st.vals), ptr(st.geom), ptr(bg), ptr(mask),
            ptr(st.final_T), ptr(st.n_contrib), ptr(g_color), ptr(g_allmap), ptr(dgeom),
            ptr(st.tile_order), ptr(stable.view(torch.uint8)), stream_ptr(dev)),
            "vings_surfel_blend_backward_accumulate_stable")
        return dgeom
    if dgeom is not None and st.tile_order is not None:
        check(L.vings_surfel_blend_backward_accumulate_ordered(
            1 if lean else 0, N, H, W, ptr(st.ranges), ptr(st.vals), ptr(st.geom), ptr(bg), ptr(mask),
            ptr(st.final_T), ptr(st.n_contrib), ptr(g_color), ptr(g_allmap), ptr(dgeom),
            ptr(st.tile_order), stream_ptr(dev)), "vings_surfel_blend_backward_accumulate_ordered")
        return dgeom
    if dgeom is not None:
        bwd = (L.vings_surfel_blend_backward_accumulate_lean if lean
               else L.vings_surfel_blend_backward_accumulate)
        check(bwd(N, H, W, ptr(st.ranges), ptr(st.vals), ptr(st.geom), ptr(bg), ptr(mask),
                  ptr(st.final_T), ptr(st.n_contrib), ptr(g_color), ptr(g_allmap), ptr(dgeom),
                  stream_ptr(dev)), "vings_surfel_blend_backward_accumulate")
        return dgeom
    if aux[4] is None:
        # no flow on the splats: the lean pass is exact whenever the incoming median-depth plane
        # is all zeros, which is decided on the device (vings_surfel_blend_backward_auto)
        dgeom = torch.zeros((max(N, 1), GRAD_STRIDE), dtype=torch.float32, device=dev)
        flag = torch.empty(1, dtype=torch.int32, device=dev)
        check(L.vings_surfel_blend_backward_auto(
            N, H, W, ptr(st.ranges), ptr(st.vals), ptr(st.geom), ptr(bg), ptr(mask), ptr(st.final_T),
            ptr(st.n_contrib), ptr(g_color), ptr(g_allmap), ptr(dgeom), ptr(st.tile_order), ptr(flag),
            stream_ptr(dev)), "vings_surfel_blend_backward_auto")
        return dgeom
    if st.tile_order is not None:
        # (the full pass in the forward pass's tile order; the ordered entry adds into its rows)
        dgeom = torch.zeros((max(N, 1), GRAD_STRIDE), dtype=torch.float32, device=dev)
        check(L.vings_surfel_blend_backward_accumulate_ordered(
            0, N, H, W, ptr(st.ranges), ptr(st.vals), ptr(st.geom), ptr(bg), ptr(mask),
            ptr(st.final_T), ptr(st.n_contrib), ptr(g_color), ptr(g_allmap), ptr(dgeom),
            ptr(st.tile_order), stream_ptr(dev)), "vings_surfel_blend_backward_accumulate_ordered")
        return dgeom
    dgeom = torch.empty((max(N, 1), GRAD_STRIDE), dtype=torch.float32, device=dev)
    check(L.vings_surfel_blend_backward(
        N, H, W, ptr(st.ranges), ptr(st.vals), ptr(st.geom), ptr(bg), ptr(mask), ptr(st.final_T),
        ptr(st.n_contrib), ptr(g_color), ptr(g_allmap), ptr(dgeom), stream_ptr(dev)),
        "vings_surfel_blend_backward")
    return dgeom


def rasterize_backward(st, aux, means3D, scales, rotations, scale_modifier, g_color, g_allmap,
                       want_means2D=True, want_scores=True):
    L = _lib.lib()
    dev = means3D.device
    view, proj, bg, mask, _ = aux
    N, H, W = st.N, st.H, st.W
    stream = stream_ptr(dev)
    dgeom = blend_backward(st, aux, g_color, g_allmap)
    d_means3D = torch.empty((N, 3), dtype=torch.float32, device=dev)
    d_scales = torch.empty((N, 2), dtype=torch.float32, device=dev)
    d_rot = torch.empty((N, 4), dtype=torch.float32, device=dev)
    d_opac = torch.empty((N, 1), dtype=torch.float32, device=dev)
    d_colors = torch.empty((N, 3), dtype=torch.float32, device=dev)
    d_means2D = torch.empty((N, 3), dtype=torch.float32, device=dev) if want_means2D else None
    d_scores = torch.empty((N, 2), dtype=torch.float32, device=dev) if want_scores else None
    check(L.vings_surfel_project_backward(
        N, H, W, float(scale_modifier), ptr(means3D), ptr(scales), ptr(rotations), ptr(view),
        ptr(proj), ptr(st.radii), ptr(dgeom), ptr(d_means3D), ptr(d_scales), ptr(d_rot),
        ptr(d_opac), ptr(d_colors), ptr(d_means2D), ptr(d_scores), stream),
        "vings_surfel_project_backward")
    return d_means3D, d_means2D, d_colors, d_opac, d_scales, d_rot, d_scores


class _RasterizeSurfels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, scores, settings):
        means3D = _f32c(means3D, "means3D")
        colors = _f32c(colors, "colors_precomp")
        opacities = _f32c(opacities, "opacities")
        scales = _f32c(scales, "scales")
        rotations = _f32c(rotations, "rotations")
        N = means3D.shape[0]
        if colors.shape != (N, 3) or scales.shape != (N, 2) or rotations.shape != (N, 4) \
                or opacities.numel() != N:
            raise _lib.VingsError(
                "expected means3D[N,3] colors[N,3] opacities[N,1] scales[N,2] rotations[N,4]")
        st, aux = rasterize_forward(means3D, colors, opacities, scales, rotations, settings,
                                    order_tiles=True)
        ctx.st, ctx.aux = st, aux
        ctx.scale_modifier = float(settings.scale_modifier)
        ctx.has_means2D = means2D is not None
        ctx.has_scores = scores is not None
        ctx.save_for_backward(means3D, scales, rotations)
        radii = st.radii
        ctx.mark_non_differentiable(radii)
        return st.color, radii, st.allmap

    @staticmethod
    def backward(ctx, g_color, g_radii, g_allmap):
        means3D, scales, rotations = ctx.saved_tensors
        st = ctx.st
        dev = means3D.device
        if g_color is None:
            g_color = torch.zeros((3, st.H, st.W), dtype=torch.float32, device=dev)
        if g_allmap is None:
            g_allmap = torch.zeros((AUX_CHANNELS, st.H, st.W), dtype=torch.float32, device=dev)
        g_color = g_color.float().contiguous()
        g_allmap = g_allmap.float().contiguous()
        d_means3D, d_means2D, d_colors, d_opac, d_scales, d_rot, d_scores = rasterize_backward(
            st, ctx.aux, means3D, scales, rotations, ctx.scale_modifier, g_color, g_allmap,
            want_means2D=ctx.has_means2D, want_scores=ctx.has_scores)
        return d_means3D, d_means2D, d_colors, d_opac, d_scales, d_rot, d_scores, None


class GaussianRasterizer(torch.nn.Module):
    """Callable with the keyword signature used at gaussian_base.py:193-203."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D=None, shs=None, colors_precomp=None, opacities=None,
                scales=None, rotations=None, scores=None, cov3D_precomp=None):
        if shs is not None or colors_precomp is None:
            raise _lib.VingsError("only colors_precomp is supported (sh_degree 0, shs=None)")
        if cov3D_precomp is not None:
            raise _lib.VingsError("cov3D_precomp is not supported; pass scales and rotations")
        if scales is None or rotations is None or opacities is None:
            raise _lib.VingsError("scales, rotations and opacities are required")
        return _RasterizeSurfels.apply(means3D, means2D, colors_precomp, opacities, scales,
                                       rotations, scores, self.raster_settings)


class SparseGaussianAdam(torch.optim.Adam):
    """Adam that only touches rows flagged visible (no bias correction).

    Constructed as ``SparseGaussianAdam(param_groups, lr=0.0, eps=1e-15)`` and
    stepped as ``step(radii > 0, N)`` (gaussian_base.py:62,402).  State keys
    ``exp_avg`` / ``exp_avg_sq`` match torch.optim.Adam because the reference
    edits them by name when splats are added or pruned (gaussian_base.py:80-110).
    """

    def __init__(self, params, lr, eps):
        super().__init__(params=params, lr=lr, eps=eps)

    def _moments(self, p):
        state = self.state[p]
        if len(state) == 0:
            state["step"] = torch.tensor(0.0, dtype=torch.float32)
            state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return state

    @torch.no_grad()
    def row_step_args(self, N, widths):
        """Host pointer tables of the parameter groups, in group order, for a kernel that
        applies this optimiser's update row by row itself (vings_splat_exchange_step):
        (params, exp_avgs, exp_avg_sqs, lrs, b1, b2, eps).  `widths` are the row widths the
        kernel assumes; all groups must share one (b1, b2, eps)."""
        groups = self.param_groups
        if len(groups) != len(widths):
            raise _lib.VingsError(f"{len(widths)} parameter groups expected, {len(groups)} found")
        keys = {(float(g["eps"]),) + tuple(float(b) for b in g.get("betas", (0.9, 0.999)))
                for g in groups}
        if len(keys) != 1:
            raise _lib.VingsError("the parameter groups differ in eps / betas")
        eps, b1, b2 = next(iter(keys))
        ps, ms, vs, lrs = [], [], [], []
        for g, width in zip(groups, widths):
            (p,) = g["params"]
            _lib.require_gpu_tensor(p, "param", torch.float32)
            if p.numel() != N * width or not p.is_contiguous():
                raise _lib.VingsError(f"group {g.get('name')}: expected a dense [{N},{width}] array")
            st = self._moments(p)
            ps.append(ptr(p)); ms.append(ptr(st["exp_avg"])); vs.append(ptr(st["exp_avg_sq"]))
            lrs.append(float(g["lr"]))
        n = len(widths)
        return ((ctypes.c_void_p * n)(*ps), (ctypes.c_void_p * n)(*ms), (ctypes.c_void_p * n)(*vs),
                (ctypes.c_float * n)(*lrs), b1, b2, eps)

    @torch.no_grad()
    def step(self, visibility, N, guard=None):
        """`guard` = (device address of a view's instance count, capacity): the step does
        nothing if the count exceeds the capacity (vings_sparse_adam_multi_guarded)."""
        L = _lib.lib()
        _lib.require_gpu_tensor(visibility, "visibility")
        vis = visibility.reshape(-1).to(torch.uint8).contiguous()
        if vis.numel() != N:
            raise _lib.VingsError("visibility must hold N entries")
        by_eps = {}
        for group in self.param_groups:
            if len(group["params"]) != 1:
                raise _lib.VingsError("more than one tensor in group")
            p = group["params"][0]
            if p.grad is None:
                continue
            _lib.require_gpu_tensor(p, "param", torch.float32)
            state = self._moments(p)
            if p.numel() % N != 0:
                raise _lib.VingsError("parameter size is not a multiple of N")
            b1, b2 = group.get("betas", (0.9, 0.999))
            key = (float(group["eps"]), float(b1), float(b2))
            by_eps.setdefault(key, []).append(
                (p, p.grad.contiguous(), state["exp_avg"], state["exp_avg_sq"],
                 p.numel() // N, float(group["lr"])))
        stream = stream_ptr(vis.device)
        for (eps, b1, b2), items in by_eps.items():
            for c0 in range(0, len(items), 8):
                chunk = items[c0:c0 + 8]
                n = len(chunk)
                if guard is not None and any(ptr(t) % 16 for c in chunk for t in c[:4]):
                    raise _lib.VingsError("the guarded Adam step needs 16-byte aligned arrays")
                if any(ptr(t) % 16 for c in chunk for t in c[:4]):
                    # the fused launch wants 16-byte aligned arrays (views into a larger
                    # buffer may not be): one launch per group instead
                    for p_, g_, m_, v_, width, lr in chunk:
                        check(L.vings_sparse_adam(ptr(p_), ptr(g_), ptr(m_), ptr(v_), ptr(vis), lr,
                                                  b1, b2, eps, N, width, stream), "vings_sparse_adam")
                    continue
                PP = (ctypes.c_void_p * n)(*[ptr(c[0]) for c in chunk])
                GG = (ctypes.c_void_p * n)(*[ptr(c[1]) for c in chunk])
                MM = (ctypes.c_void_p * n)(*[ptr(c[2]) for c in chunk])
                VV = (ctypes.c_void_p * n)(*[ptr(c[3]) for c in chunk])
                WW = (ctypes.c_int * n)(*[c[4] for c in chunk])
                LR = (ctypes.c_float * n)(*[c[5] for c in chunk])
                if guard is not None:
                    check(L.vings_sparse_adam_multi_guarded(
                        n, PP, GG, MM, VV, WW, LR, ptr(vis), b1, b2, eps, N, guard[0], int(guard[1]),
                        stream), "vings_sparse_adam_multi_guarded")
                    continue
                check(L.vings_sparse_adam_multi(n, PP, GG, MM, VV, WW, LR, ptr(vis), b1, b2, eps,
                                                N, stream), "vings_sparse_adam_multi")
