"""Python wrapper around the rasteriser that the mapping loop needs: render
post-processing, the mapping loss and the score-weighted gradient step.

This is the counterpart of the reference's own Python around its native
extension (it stays PyTorch there too):
  render post-processing  /root/reference/scripts/gaussian/gaussian_base.py:142-224
  depth -> normal         /root/reference/scripts/gaussian/normal_utils.py:6-21
                          /root/reference/scripts/gaussian/tf.py:133-142
  loss                    /root/reference/scripts/gaussian/loss_utils.py:7-153
  score weighting         /root/reference/scripts/gaussian/gaussian_utils.py:56-62
  hot loop                /root/reference/scripts/gaussian/gaussian_base.py:343-404
The formulas are pinned by fixtures generated from those reference functions
(tests/golden/harness_golden.npz).  Masked means are written as weighted sums
instead of boolean indexing so the loop never synchronises with the host.
"""
import math

import torch
import torch.nn.functional as F

from . import synth
from .surfel import GaussianRasterizationSettings, GaussianRasterizer, SparseGaussianAdam

# training_args of /root/reference/configs/hierarchical/smallcity.yaml:67-81
DEFAULT_LR = {"_xyz_lr": 0.00016, "_rgb_lr": 0.0025, "_opacity_lr": 0.05, "_scaling_lr": 0.005,
              "_rotation_lr": 0.001}
DEFAULT_LOSS_WEIGHTS = {"rgb_loss": 1.0, "depth_loss": 1.0, "alpha_loss": 1.0,
                        "normal_loss": 0.1, "dist_loss": 0.0}


def backproject_depth(depth, intr):
    """(H,W) depth -> (H,W,3) camera-space points; u = row (vertical), v = column."""
    H, W = depth.shape
    u = torch.arange(H, device=depth.device, dtype=depth.dtype).view(H, 1)
    v = torch.arange(W, device=depth.device, dtype=depth.dtype).view(1, W)
    x = (v - intr["cv"]) / intr["fv"] * depth
    y = (u - intr["cu"]) / intr["fu"] * depth
    return torch.stack((x, y, depth), dim=-1)


def depth_to_normal(depth, intr):
    """Camera-space surface normal from a depth map by central differences; zero border."""
    pts = backproject_depth(depth, intr)
    dx = pts[1:-1, 2:] - pts[1:-1, :-2]
    dy = pts[:-2, 1:-1] - pts[2:, 1:-1]
    n = F.normalize(torch.cross(dx, dy, dim=-1), p=2, dim=-1)
    return F.pad(n, (0, 0, 1, 1, 1, 1))


def render(params, w2c, intr, bg=None, pixel_mask=None, u2_minus_u1=None, scores=None,
           means2D=None):
    """Render one view; returns the dict consumed by the loss (gaussian_base.py:211-224)."""
    rgb, radii, allmap = render_raw(params, w2c, intr, bg, pixel_mask, u2_minus_u1, scores,
                                    means2D)
    return postprocess(rgb, radii, allmap, w2c, intr)


def view_settings(w2c, intr, bg=None, pixel_mask=None, u2_minus_u1=None):
    """The settings record the reference builds per render (gaussian_base.py:160-177) for the
    camera `w2c` (device tensor).  A keyframe's record can be kept and reused."""
    cam = synth.Camera(w2c, intr)
    if bg is None:
        bg = torch.zeros(3, device=w2c.device)
    return GaussianRasterizationSettings(
        image_height=cam.height, image_width=cam.width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=0, campos=cam.camera_center,
        prefiltered=False, debug=False, pixel_mask=pixel_mask, u2_minus_u1=u2_minus_u1)


def render_raw(params, w2c, intr, bg=None, pixel_mask=None, u2_minus_u1=None, scores=None,
               means2D=None):
    """Render one view; returns the rasteriser's own outputs (rgb, radii, allmap)."""
    settings = view_settings(w2c, intr, bg, pixel_mask, u2_minus_u1)
    act = synth.activate(params)
    return GaussianRasterizer(raster_settings=settings)(
        means3D=act["means3D"], means2D=means2D, shs=None, colors_precomp=act["colors"],
        opacities=act["opacities"], scales=act["scales"], rotations=act["rotations"],
        scores=scores, cov3D_precomp=None)


def postprocess(rgb, radii, allmap, w2c, intr):
    alpha = allmap[1:2]
    depth = torch.nan_to_num(allmap[0:1] / alpha, 0, 0)
    Rw2c = w2c[:3, :3]
    normal = (allmap[2:5].permute(1, 2, 0) @ Rw2c).permute(2, 0, 1)
    surf = depth_to_normal(depth.squeeze(0), intr)
    surf = (surf @ Rw2c).permute(2, 0, 1)
    return {"radii": radii, "accum": alpha, "rgb": rgb, "depth": depth, "normal": normal,
            "dist": allmap[6:7], "surf_normal": surf, "flow": allmap[7:9]}


_WINDOWS = {}


def _ssim_window(channel, device, dtype):
    key = (channel, str(device), dtype)
    if key not in _WINDOWS:
        g = torch.tensor([math.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)])
        g = (g / g.sum()).unsqueeze(1)
        w2 = (g @ g.t()).float()[None, None].expand(channel, 1, 11, 11).contiguous()
        _WINDOWS[key] = w2.to(device=device, dtype=dtype)
    return _WINDOWS[key]


def ssim_map(img1, img2):
    c = img1.shape[-3]
    w = _ssim_window(c, img1.device, img1.dtype)
    mu1 = F.conv2d(img1, w, padding=5, groups=c)
    mu2 = F.conv2d(img2, w, padding=5, groups=c)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=5, groups=c) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=5, groups=c) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=5, groups=c) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))


def _masked_mean(x, mask_f, count):
    # mean over the selected pixels of all channels of x ([C,H,W])
    return (x * mask_f).sum() / (count * x.shape[0])


def mapping_loss(pred, gt, weights=DEFAULT_LOSS_WEIGHTS):
    """Total mapping loss (use_sky = False branch of loss_utils.get_loss)."""
    sky = gt["rgb"].sum(dim=0) == 0.0
    valid = (~sky) & (gt["depth"].sum(dim=0) > 0.0)
    vf, sf = valid.to(pred["rgb"].dtype), sky.to(pred["rgb"].dtype)
    nv, ns = vf.sum().clamp(min=1.0), sf.sum().clamp(min=1.0)
    l1 = _masked_mean((pred["rgb"] - gt["rgb"]).abs(), vf, nv)
    ssim = _masked_mean(ssim_map(pred["rgb"], gt["rgb"]), vf, nv)
    rgb_loss = 0.8 * l1 + 0.2 * (1.0 - ssim)
    normal_loss = ((1 - (pred["normal"] * pred["surf_normal"]).sum(dim=0)) * vf).sum() / nv
    alpha_loss = _masked_mean(pred["accum"], sf, ns)
    depth_loss = _masked_mean((pred["depth"] - gt["depth"]).abs() / gt["depth_cov"], vf, nv)
    dist_loss = _masked_mean(pred["dist"], vf, nv)
    return (weights["rgb_loss"] * rgb_loss + weights["normal_loss"] * normal_loss +
            weights["alpha_loss"] * alpha_loss + weights["depth_loss"] * depth_loss +
            weights["dist_loss"] * dist_loss)


_LOSS_WS = {}
LOSS_WEIGHT_ORDER = ("rgb_loss", "depth_loss", "alpha_loss", "normal_loss", "dist_loss")


def mapping_loss_hip(rgb, allmap, gt, Rw2c_host, intr, weights=DEFAULT_LOSS_WEIGHTS, cache=False):
    """`mapping_loss(postprocess(...))` and its gradient w.r.t. the rasteriser outputs in
    four HIP launches (csrc/mapping_loss.hip) instead of a PyTorch graph.

    rgb [3,H,W], allmap [9,H,W]: rasteriser outputs (device); gt: dict with rgb [3,H,W],
    depth [1,H,W], depth_cov [1,H,W]; Rw2c_host: 3x3 CPU tensor (rotation of w2c).
    Returns (loss_vec [8] device tensor, dL_drgb, dL_dallmap); loss_vec[0] is the total.

    cache=True keeps what depends on the ground truth alone (filtered moments of gt["rgb"],
    pixel classes, their counts) in gt["_loss_cache"] between calls, which is what the mapper
    wants while it iterates over a keyframe.  The entry is tied to the two image tensors
    themselves: it holds references to them (so their addresses cannot be handed to other
    tensors while it lives) and remembers their version counters (so an in-place torch write
    invalidates it); new tensors assigned to gt["rgb"] / gt["depth"] never match.  Only a write
    through a raw pointer (a kernel of another library) escapes it: then drop the entry with
    `gt.pop("_loss_cache", None)`.  Same bits with and without.
    """
    from . import _lib
    import ctypes
    lib = _lib.lib()
    dev = rgb.device
    _lib.require_gpu_tensor(rgb, "rgb", torch.float32)
    _lib.require_gpu_tensor(allmap, "allmap", torch.float32)
    H, W = rgb.shape[-2:]
    key = (H, W, dev.index)
    if key not in _LOSS_WS:
        nbytes = lib.vings_mapping_loss_workspace_bytes(H, W)
        _LOSS_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ws = _LOSS_WS[key]
    d_rgb = torch.empty_like(rgb)
    d_allmap = torch.empty_like(allmap)
    loss_vec = torch.empty(8, dtype=torch.float32, device=dev)
    R = (ctypes.c_float * 9)(*[float(x) for x in Rw2c_host.reshape(-1).tolist()])
    wts = (ctypes.c_float * 5)(*[float(weights[k]) for k in LOSS_WEIGHT_ORDER])
    gt_rgb, gt_depth, gt_cov = (gt["rgb"].contiguous(), gt["depth"].contiguous(),
                                gt["depth_cov"].contiguous())
    args = (H, W, _lib.ptr(rgb.contiguous()), _lib.ptr(allmap.contiguous()), _lib.ptr(gt_rgb),
            _lib.ptr(gt_depth), _lib.ptr(gt_cov), R, float(intr["fu"]), float(intr["fv"]),
            float(intr["cu"]), float(intr["cv"]), wts, _lib.ptr(d_rgb), _lib.ptr(d_allmap),
            _lib.ptr(loss_vec), _lib.ptr(ws), ws.numel())
    if cache:
        tag = (gt_rgb.data_ptr(), gt_depth.data_ptr(), H, W, gt_rgb._version, gt_depth._version)
        ent = gt.get("_loss_cache")
        valid = ent is not None and ent[0] == tag and ent[2] is gt_rgb and ent[3] is gt_depth
        if not valid:
            # (the buffer of a stale entry is reused when the size fits: a captured view step
            # has its address baked in, see SurfelMapper._graph_signature)
            nbytes = int(lib.vings_mapping_loss_gt_cache_bytes(H, W))
            buf = ent[1] if ent is not None and ent[1].numel() == nbytes and ent[1].device == dev \
                else torch.empty(nbytes, dtype=torch.uint8, device=dev)
            ent = (tag, buf, gt_rgb, gt_depth)
            gt["_loss_cache"] = ent
        _lib.check(lib.vings_mapping_loss_cached(*args, _lib.ptr(ent[1]), 1 if valid else 0,
                                                 _lib.stream_ptr(dev)), "vings_mapping_loss_cached")
    else:
        _lib.check(lib.vings_mapping_loss(*args, _lib.stream_ptr(dev)), "vings_mapping_loss")
    return loss_vec, d_rgb, d_allmap


class _MappingLossFunction(torch.autograd.Function):
    """`mapping_loss_hip` as a differentiable scalar: forward runs the fused kernels and keeps
    the two gradient maps, backward scales them by the upstream gradient."""

    @staticmethod
    def forward(ctx, rgb, allmap, gt, Rw2c_host, intr, weights, cache):
        loss_vec, d_rgb, d_allmap = mapping_loss_hip(rgb.detach(), allmap.detach(), gt, Rw2c_host,
                                                     intr, weights, cache=cache)
        ctx.save_for_backward(d_rgb, d_allmap)
        ctx.mark_non_differentiable(loss_vec)
        return loss_vec[0].clone(), loss_vec

    @staticmethod
    def backward(ctx, g_loss, _g_vec):
        d_rgb, d_allmap = ctx.saved_tensors
        return d_rgb * g_loss, d_allmap * g_loss, None, None, None, None, None


def mapping_loss_fused(rgb, allmap, gt, Rw2c_host, intr, weights=DEFAULT_LOSS_WEIGHTS, cache=False):
    """The mapping loss of a rendered view as a scalar tensor that `backward()` can be called
    on, computed by the fused HIP kernels: for a training loop written like the reference's
    (gaussian_base.py:367-382: render, `get_loss`, `loss.backward()`) that keeps the
    rasteriser's raw outputs `rgb` [3,H,W] and `allmap` [9,H,W].  Returns (loss, loss_vec):
    loss_vec [8] holds total, rgb, normal, alpha, depth, dist terms and the two pixel counts."""
    return _MappingLossFunction.apply(rgb, allmap, gt, Rw2c_host, intr, weights, cache)


def score_weight(current_scores, global_scores):
    """Per-splat gradient weight s0 / (S0 + 1e-6 + s0), [N,1]."""
    s0 = current_scores[:, 0]
    return (s0 / (global_scores[:, 0] + 1e-6 + s0)).unsqueeze(1)


class SurfelMapper:
    """Splat parameter store + the per-view optimisation step of the mapping loop."""

    NAMES = ("_xyz", "_rgb", "_opacity", "_scaling", "_rotation")

    def __init__(self, raw_params, device, lr=DEFAULT_LR, loss_weights=DEFAULT_LOSS_WEIGHTS):
        self.device = torch.device(device)
        self.params = {k: torch.nn.Parameter(raw_params[k].to(self.device).float().contiguous())
                       for k in self.NAMES}
        n = self.params["_xyz"].shape[0]
        self.n = n
        self.loss_weights = dict(loss_weights)
        groups = [{"params": [self.params[k]], "lr": lr[k + "_lr"], "name": k} for k in self.NAMES]
        self.optimizer = SparseGaussianAdam(groups, lr=0.0, eps=1e-15)
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=self.device)
        self.global_scores = z(n, 2)
        self.local_scores = z(n, 2)
        self.stable_mask = z(n, dt=torch.bool)
        self.globalkf_id = z(n, dt=torch.int64)
        self.globalkf_max_scores = z(n)
        self.last_loss = None
        # True once the stable mask holds any splat (maintenance.stablemask_control keeps it; a
        # caller that edits stable_mask by hand calls note_stable_mask): the fused view step
        # then takes the backward pass that skips the gradients of stable splats.  Only a
        # choice between two kernels that give the optimiser the same rows.
        self.stable_skip = False

    def note_stable_mask(self):
        """Call after editing `stable_mask` directly (one host read of a device flag)."""
        self.stable_skip = bool(self.stable_mask.any())
        return self.stable_skip

    def render(self, w2c, intr, scores=None):
        return render(self.params, w2c, intr, scores=scores)

    def backward_view(self, w2c, intr, gt, Rw2c_host=None, torch_loss=False):
        """Forward + loss + backward for one view; leaves .grad on the parameters.
        Returns (radii, current_scores).

        The loss and its gradient w.r.t. the rendered maps come from the fused HIP kernels
        (`mapping_loss_hip`); `torch_loss=True` builds the PyTorch graph of `mapping_loss`
        instead (the formulation the fixtures pin, kept as the cross-check).  `Rw2c_host` is
        the CPU copy of w2c[:3,:3]; passing it avoids a device->host read per view."""
        for k in self.NAMES:
            # (a fused view step leaves its own gradient buffers bound to .grad; autograd would
            # add this view's gradients to whatever they hold)
            self.params[k].grad = None
        scores = torch.zeros((self.n, 2), device=self.device, requires_grad=True)
        if torch_loss:
            pred = self.render(w2c, intr, scores=scores)
            loss = mapping_loss(pred, gt, self.loss_weights)
            loss.backward()
            self.last_loss = loss.detach()
            return pred["radii"], scores.grad.detach()
        rgb, radii, allmap = render_raw(self.params, w2c, intr, scores=scores)
        if Rw2c_host is None:
            Rw2c_host = w2c[:3, :3].cpu()
        loss, loss_vec = mapping_loss_fused(rgb, allmap, gt, Rw2c_host, intr, self.loss_weights,
                                            cache=True)
        loss.backward()
        self.last_loss = loss_vec[0]
        self.last_loss_terms = loss_vec
        return radii, scores.grad.detach()

    def finish_view(self, radii, cur, kf_id, reducer=None):
        """Score bookkeeping, gradient weighting, (multi-GPU) gradient exchange and the
        sparse Adam step for one view (gaussian_base.py:385-403,
        gaussian_model.py:163-168).  With one rank this is exactly the reference order:
        record the scores, weight every gradient by s0 / (S0 + 1e-6 + s0) using the updated
        global score, zero the radii of stable splats, step."""
        self._match_reducer(reducer)
        s0 = cur[:, 0]
        g0_local = (self.global_scores[:, 0] + s0).clamp(0.0, 1e4)
        w = (s0 / (g0_local + 1e-6 + s0)).unsqueeze(1)
        grads = {k: self.params[k].grad for k in self.NAMES}
        for k in self.NAMES:
            grads[k].mul_(w)
        vis = (radii > 0) & (~self.stable_mask)
        if reducer is not None and reducer.world > 1:
            from . import parallel
            vis, s0_sum = reducer.reduce(grads, cur, vis)
            self._pending_keys = torch.maximum(self._record_keys(), parallel.record_keys(cur, kf_id))
            self._records_dirty = True
            self.local_scores[:, 0] += s0_sum
            self.global_scores[:, 0] = (self.global_scores[:, 0] + s0_sum).clamp(0.0, 1e4)
        else:
            self.local_scores[:, 0] += s0
            self.global_scores[:, 0] = (self.global_scores[:, 0] + s0).clamp(0.0, 1e4)
            self.local_scores[:, 1] = torch.maximum(self.local_scores[:, 1], cur[:, 1])
            replace = self.globalkf_max_scores < s0
            self.globalkf_max_scores = torch.where(replace, s0, self.globalkf_max_scores)
            self.globalkf_id = torch.where(replace, torch.full_like(self.globalkf_id, kf_id),
                                           self.globalkf_id)
        self.optimizer.step(vis, self.n)
        # (gaussian_base.py:403 calls zero_grad() with torch's default, which drops the gradient
        # tensors: the next backward assigns fresh ones instead of filling and adding)
        self.optimizer.zero_grad()
        self._last_radii = radii

    # ------------------------------------------------------------------ multi-GPU records
    def _match_reducer(self, reducer):
        """Map maintenance (prune, add_splats) changes the splat count between view steps; the
        exchange buffer and the pending record keys must follow, or the packing kernels index
        them with the wrong N."""
        if reducer is not None and reducer.world > 1 and reducer.n != self.n:
            reducer.resize(self.n)
        pending = getattr(self, "_pending_keys", None)
        if pending is not None and pending.shape[0] != self.n:
            from . import _lib
            raise _lib.VingsError(
                f"pending record keys hold {pending.shape[0]} rows for {self.n} splats: change the "
                "splat count through maintenance.prune / add_splats, which carry them along")

    def _record_keys(self):
        """Running maximum of this rank's (best score | keyframe, error score) keys since the
        last sync_records; int64 [N,2]."""
        if getattr(self, "_pending_keys", None) is None:
            self._pending_keys = torch.zeros((self.n, 2), dtype=torch.int64, device=self.device)
        self._records_dirty = True       # a view step is about to fold its scores in
        return self._pending_keys

    def sync_records(self, reducer, async_op=False):
        """Folds the record maxima the ranks gathered since the last call into
        local_scores[:,1], globalkf_max_scores and globalkf_id (one MAX all-reduce).  The view
        steps do not read these three, so a multi-GPU job calls this once per window and before
        anything that does (storage control, loop closure, checkpoints); with one rank the
        records are always current and this does nothing.  With async_op the collective is only
        started and a function that completes the merge is returned (None when there is nothing
        to do), so independent work can be queued in between."""
        if reducer is None or reducer.world == 1 or getattr(self, "_pending_keys", None) is None \
                or not getattr(self, "_records_dirty", True):
            return None
        from . import _lib
        keys = self._pending_keys
        work = reducer.all_reduce_record_keys(keys, async_op=async_op)

        def finish():
            if async_op:
                work.wait()
            _lib.check(_lib.lib().vings_splat_merge_records(
                self.n, _lib.ptr(keys), _lib.ptr(self.local_scores),
                _lib.ptr(self.globalkf_max_scores), _lib.ptr(self.globalkf_id),
                _lib.stream_ptr(self.device)), "vings_splat_merge_records")
            self._records_dirty = False  # nothing unmerged until the next view step

        if async_op:
            return finish
        finish()
        return None

    # ------------------------------------------------------------------ fused view step
    def _fused_buffers(self):
        if getattr(self, "_fb", None) is None:
            n, dev = self.n, self.device
            e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)
            fb = {"opac": e(n, 1), "scales": e(n, 2), "rot": e(n, 4), "vis": e(n, dt=torch.uint8),
                  "grads": {k: torch.zeros_like(self.params[k]) for k in self.NAMES}}
            self._fb = fb
        for k in self.NAMES:
            if self.params[k].grad is not self._fb["grads"][k]:
                self.params[k].grad = self._fb["grads"][k]   # the optimiser reads .grad
        return self._fb

    def fused_view(self, settings, intr, gt, Rw2c_host, kf_id=0, reducer=None, lean=None):
        """The same view step as backward_view + finish_view with no PyTorch op in it:
        activations, rasteriser forward, loss, rasteriser backward, chain rule + score
        weighting + bookkeeping, sparse Adam — about 22 HIP launches through the C ABI.
        `settings` is the view's GaussianRasterizationSettings (see `view_settings`).
        `lean=False` forces the full blend passes (median depth and flow planes, which the
        mapping loss never reads) where the default takes the lean ones."""
        from . import _lib, surfel
        L = _lib.lib()
        self._match_reducer(reducer)
        fb = self._fused_buffers()
        P, n, dev = self.params, self.n, self.device
        stream = _lib.stream_ptr(dev)
        ptr = _lib.ptr
        with torch.no_grad():
            # the projection applies the activations itself and fills fb["opac" / "scales" / "rot"]
            xyz, rgb = P["_xyz"].detach(), P["_rgb"].detach()
            dgeom = torch.empty((max(n, 1), surfel.GRAD_STRIDE), dtype=torch.float32, device=dev)
            # (the loss has no term in the median depth or the flow: the lean blend passes)
            lean = settings.u2_minus_u1 is None and (lean is None or lean)
            st, aux = surfel.rasterize_forward(
                xyz, rgb, fb["opac"], fb["scales"], fb["rot"], settings,
                raw=(P["_opacity"].detach(), P["_scaling"].detach(), P["_rotation"].detach()),
                dgeom=dgeom, lean=lean)
            single = reducer is None or reducer.world == 1
            compact = not single and reducer.compact
            if compact:
                # which splats does some rank touch in this step?  Known from the projection
                # on; agreed between the ranks on a side stream while this one goes on
                reducer.start_union(st.radii)
            loss_vec, d_rgb, d_allmap = mapping_loss_hip(st.color, st.allmap, gt, Rw2c_host, intr,
                                                         self.loss_weights, cache=True)
            g = fb["grads"]
            view, proj = aux[0], aux[1]
            surfel.blend_backward(st, aux, d_rgb, d_allmap, dgeom=dgeom, lean=lean,
                                  stable=self.stable_mask if self.stable_skip else None)
            if compact:
                self._exchange_compact(L, reducer, st, xyz, fb, view, proj, dgeom, kf_id, stream)
            else:
                # per-splat backward + chain rule through the activations + score weighting +
                # bookkeeping in one launch
                _lib.check(L.vings_surfel_project_backward_finish(
                    n, st.H, st.W, 1.0, ptr(xyz), ptr(fb["scales"]), ptr(fb["rot"]), ptr(view),
                    ptr(proj), ptr(st.radii), ptr(dgeom), ptr(fb["opac"]), ptr(P["_rotation"]),
                    ptr(self.stable_mask.view(torch.uint8)), ptr(self.global_scores),
                    ptr(self.local_scores), ptr(self.globalkf_max_scores), ptr(self.globalkf_id),
                    int(kf_id), 1 if single else 0, ptr(g["_xyz"]), ptr(g["_rgb"]),
                    ptr(g["_opacity"]), ptr(g["_scaling"]), ptr(g["_rotation"]), ptr(fb["vis"]), None,
                    None if single else ptr(reducer.fbuf),
                    None if single else ptr(self._record_keys()), stream),
                    "vings_surfel_project_backward_finish")
                vis = fb["vis"]
                if not single:
                    # (packed by the kernel above) -> SUM all-reduce over the replicas -> unpack
                    grad_ptrs = [ptr(g[k]) for k in self.NAMES]
                    reducer.all_reduce_packed()
                    _lib.check(L.vings_splat_unpack_exchange(
                        n, ptr(reducer.fbuf), ptr(self.global_scores), ptr(self.local_scores),
                        *grad_ptrs, ptr(vis), stream), "vings_splat_unpack_exchange")
                self.optimizer.step(vis, n)
        self.last_loss = loss_vec[0]
        self.last_loss_terms = loss_vec
        self._last_radii = st.radii
        self._last_instances = int(st.D)
        return loss_vec

    # ------------------------------------------------------------------ view step as a HIP graph
    def _fused_view_fixed(self, settings, intr, gt, Rw2c_host, kf_id, capacity, count_words):
        """`fused_view` (single GPU) in the form a stream capture can record: fixed instance
        capacity, no host wait, and a finish + Adam that skip the whole step on the device if
        the view's instance count exceeds the capacity."""
        from . import _lib, surfel
        L = _lib.lib()
        fb = self._fused_buffers()
        P, n, dev = self.params, self.n, self.device
        stream = _lib.stream_ptr(dev)
        ptr = _lib.ptr
        with torch.no_grad():
            xyz, rgb = P["_xyz"].detach(), P["_rgb"].detach()
            dgeom = torch.empty((n, surfel.GRAD_STRIDE), dtype=torch.float32, device=dev)
            st, aux = surfel.rasterize_forward_fixed(
                xyz, rgb, fb["opac"], fb["scales"], fb["rot"], settings,
                (P["_opacity"].detach(), P["_scaling"].detach(), P["_rotation"].detach()), dgeom,
                capacity, count_words)
            loss_vec, d_rgb, d_allmap = mapping_loss_hip(st.color, st.allmap, gt, Rw2c_host, intr,
                                                         self.loss_weights, cache=True)
            g = fb["grads"]
            view, proj = aux[0], aux[1]
            surfel.blend_backward(st, aux, d_rgb, d_allmap, dgeom=dgeom,
                                  lean=settings.u2_minus_u1 is None,
                                  stable=self.stable_mask if self.stable_skip else None)
            count_ptr = st.tail_ptr + 8                    # third count word: D
            _lib.check(L.vings_surfel_project_backward_finish_guarded(
                n, st.H, st.W, 1.0, ptr(xyz), ptr(fb["scales"]), ptr(fb["rot"]), ptr(view),
                ptr(proj), ptr(st.radii), ptr(dgeom), ptr(fb["opac"]), ptr(P["_rotation"]),
                ptr(self.stable_mask.view(torch.uint8)), ptr(self.global_scores),
                ptr(self.local_scores), ptr(self.globalkf_max_scores), ptr(self.globalkf_id),
                int(kf_id), ptr(g["_xyz"]), ptr(g["_rgb"]), ptr(g["_opacity"]), ptr(g["_scaling"]),
                ptr(g["_rotation"]), ptr(fb["vis"]), count_ptr, int(capacity), stream),
                "vings_surfel_project_backward_finish_guarded")
            self.optimizer.step(fb["vis"], n, guard=(count_ptr, capacity))
        return loss_vec, st

    def drop_view_graphs(self):
        """Forget the captured view steps (and give their memory pool back): the arrays they
        address are about to be replaced."""
        self.__dict__.pop("_view_graphs", None)
        self.__dict__.pop("_view_graph_pool", None)

    def _graph_signature(self, settings, gt):
        """Everything a captured view step has baked in by address: a change (map maintenance
        replaces the parameter and moment arrays, a new keyframe brings new images) makes the
        graph stale."""
        sig = [self.n, self.stable_skip]
        for k in self.NAMES:
            p = self.params[k]
            st = self.optimizer.state.get(p, {})
            sig += [p.data_ptr(), st["exp_avg"].data_ptr() if "exp_avg" in st else 0]
        sig += [t.data_ptr() for t in (self.global_scores, self.local_scores, self.stable_mask,
                                       self.globalkf_id, self.globalkf_max_scores)]
        sig += [gt[k].data_ptr() for k in ("rgb", "depth", "depth_cov")]
        # the loss kernels' per-keyframe cache: its address is baked in, and so is "the cache
        # is valid", which only holds while the images are the tensors it was filled from
        ent = gt.get("_loss_cache")
        sig += [0 if ent is None else ent[1].data_ptr(), gt["rgb"]._version, gt["depth"]._version,
                id(gt["rgb"]), id(gt["depth"])]
        sig += [settings.viewmatrix.data_ptr(), settings.projmatrix.data_ptr()]
        sig += [float(g["lr"]) for g in self.optimizer.param_groups]
        return tuple(sig)

    def fused_view_graphed(self, key, settings, intr, gt, Rw2c_host, kf_id=0):
        """`fused_view` of keyframe `key`, replayed from a HIP graph (single GPU).  The mapper
        optimises the keyframes of its window over and over (gaussian_base.py:353-404 runs
        once per iteration per keyframe); the ~25 launches of a view step differ between two
        iterations of the same keyframe in nothing but the data, so they are captured once per
        keyframe and replayed, which removes the launch gaps and the host work between them.

        The first call for a key runs the plain step (and learns the instance count), the second
        captures the step with room for 1/8 more instances, later calls replay.  `settings`,
        `gt` and the parameter arrays are baked in by address: pass the same tensors every time
        (a changed address, splat count or learning rate re-captures).  Should a replayed view
        ever produce more instances than its graph has room for, that replay leaves the map
        untouched (the guarded kernels drop the step); it is noticed at the next call for the
        key, which repeats the dropped step eagerly before it re-captures.
        Returns the loss vector of the step (device tensor; of the repeated step in that case)."""
        graphs = self.__dict__.setdefault("_view_graphs", {})
        sig = self._graph_signature(settings, gt)
        ent = graphs.get(key)
        retired = None
        if ent is not None and ent.get("sig") != sig:
            # (a retired graph is kept until its successor is captured: the graphs of a mapper
            # share one memory pool, which must not lose its last user in between)
            retired, ent = ent.get("graph") or ent.get("retired"), None
        if ent is None:
            loss_vec = self.fused_view(settings, intr, gt, Rw2c_host, kf_id)
            graphs[key] = {"sig": self._graph_signature(settings, gt), "graph": None,
                           "retired": retired, "D": self._last_instances}
            return loss_vec
        if ent["graph"] is not None:
            # the replay of one window ago has long finished; its count tells whether it fit
            ent["event"].synchronize()
            d_prev = int(ent["count"][2])
            if d_prev > ent["capacity"]:
                self.fused_view(settings, intr, gt, Rw2c_host, kf_id)     # the dropped step
                ent["retired"], ent["graph"] = ent["graph"], None
                ent["D"] = self._last_instances
                ent["sig"] = self._graph_signature(settings, gt)
            elif d_prev > 0.97 * ent["capacity"]:
                ent["retired"], ent["graph"] = ent["graph"], None         # growing: more room
                ent["D"] = d_prev
        if ent["graph"] is None:
            cap = ent["D"] + ent["D"] // 8 + 4096
            count = torch.zeros(4, dtype=torch.int32).pin_memory()
            pool = self.__dict__.get("_view_graph_pool")
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(g, pool=pool):
                loss_vec, st = self._fused_view_fixed(settings, intr, gt, Rw2c_host, kf_id, cap, count)
            if pool is None:
                self._view_graph_pool = g.pool()       # the graphs of a window share their memory
            ent.update(graph=g, capacity=cap, count=count, loss=loss_vec, radii=st.radii,
                       event=torch.cuda.Event(), retired=None)
        ent["graph"].replay()
        ent["event"].record()
        self.last_loss = ent["loss"][0]
        self.last_loss_terms = ent["loss"]
        self._last_radii = ent["radii"]
        return ent["loss"]

    _WIDTHS = (3, 3, 1, 2, 4)

    def _exchange_compact(self, L, reducer, st, xyz, fb, view, proj, dgeom, kf_id, stream):
        """The multi-GPU tail of the fused view step in its compact form: the per-splat
        backward writes the weighted gradients and the importance score of the rows some rank
        touched into the chunked exchange buffer; every chunk is summed over the ranks and
        then applied (score records + Adam on its rows) while the next one is being reduced."""
        from . import _lib
        ptr, n, P = _lib.ptr, self.n, self.params
        m, chunk_rows = reducer.finish_union()
        if m == 0:
            return
        cx = reducer._cx
        _lib.check(L.vings_surfel_project_backward_pack(
            n, st.H, st.W, 1.0, ptr(xyz), ptr(fb["scales"]), ptr(fb["rot"]), ptr(view), ptr(proj),
            ptr(st.radii), ptr(dgeom), ptr(fb["opac"]), ptr(P["_rotation"]),
            ptr(self.global_scores), int(kf_id), None, ptr(cx["packed"]),
            ptr(self._record_keys()), ptr(cx["slot"]), ptr(cx["rows"]), chunk_rows, stream),
            "vings_surfel_project_backward_pack")
        PP, MM, VV, LR, b1, b2, eps = self.optimizer.row_step_args(n, self._WIDTHS)
        stable = ptr(self.stable_mask.view(torch.uint8))
        rows_base, gs, ls = cx["rows"].data_ptr(), ptr(self.global_scores), ptr(self.local_scores)

        def apply(chunk, first_slot, n_slots):
            _lib.check(L.vings_splat_exchange_step(
                n_slots, chunk_rows, ptr(chunk), rows_base + 4 * first_slot, PP, MM, VV, LR,
                b1, b2, eps, stable, gs, ls, stream), "vings_splat_exchange_step")

        reducer.exchange_chunks(m, chunk_rows, apply)

    def train_view(self, c2w, intr, gt, kf_id=0, reducer=None):
        """One iteration of the reference's hot loop (gaussian_base.py:353-404)."""
        w2c = torch.linalg.inv(c2w)
        radii, cur = self.backward_view(w2c, intr, gt)
        self.finish_view(radii, cur, kf_id, reducer)
        return self.last_loss

    def last_stats(self):
        """Instance count D and visible-splat count of the most recent forward pass."""
        from . import surfel
        n_vis = int((self._last_radii > 0).sum().item()) if getattr(self, "_last_radii", None) \
            is not None else 0
        return {"D": int(surfel.LAST_INSTANCE_COUNT), "n_vis": n_vis}
