"""Host-side mirror of the reference's ``droid_backends`` extension: ``ba``,
``BACore``, ``frame_distance``, ``corr_index_forward`` / ``corr_index_backward``,
``altcorr_forward`` / ``altcorr_backward``.

Same call signatures as used at
/root/reference/scripts/frontend/depth_video.py:289-299,380-381,475-481,554-568,618,650
and /root/reference/scripts/frontend/modules/corr.py:12,19,79,86-87.  Compute happens in
libvings_hip.so through the C ABI (include/vings_hip.h); torch provides device
memory and the current stream.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

_c = ctypes


class _Layout(ctypes.Structure):
    _fields_ = [(n, _c.c_size_t) for n in (
        "total_bytes", "off_E", "off_Q", "off_w", "off_H", "off_v", "off_S", "off_u", "off_dx",
        "off_chol", "off_kx", "off_kk", "off_kstart", "off_kedges", "off_ip", "off_jp",
        "off_ent_start", "off_ent_row", "off_ent_pose", "off_status")]


_P = _c.c_void_p
_I = _c.c_int
_bound = {}


def _fn(name):
    if name not in _bound:
        sigs = {
            "vings_dba_layout_query": (_I, [_I] * 5 + [_c.POINTER(_Layout)]),
            "vings_dba_ba": (_I, [_P] * 9 + [_I] * 8 + [_c.c_float, _c.c_float, _I, _P,
                                                       _c.c_size_t, _P]),
            "vings_dba_linearize": (_I, [_P] * 9 + [_I] * 8 + [_P, _c.c_size_t, _P]),
            "vings_dba_reduced_system": (_I, [_I] * 7 + [_P, _c.c_size_t, _P, _P, _P]),
            "vings_dba_retract": (_I, [_P] * 3 + [_I] * 8 + [_P, _c.c_size_t, _P]),
            "vings_dba_export": (_I, [_I] * 7 + [_P, _c.c_size_t, _P, _P, _P]),
            "vings_frame_distance": (_I, [_P] * 5 + [_I] * 4 + [_c.c_float, _P, _P]),
            "vings_corr_index_forward": (_I, [_P] * 3 + [_I] * 7 + [_P]),
            "vings_corr_index_backward": (_I, [_P] * 3 + [_I] * 7 + [_P]),
            "vings_altcorr_forward": (_I, [_P] * 4 + [_I] * 8 + [_P]),
            "vings_altcorr_backward": (_I, [_P] * 6 + [_I] * 8 + [_P]),
        }
        res, args = sigs[name]
        _bound[name] = _lib.bind(name, res, args)
    return _bound[name]


def _gpu_f32(t, name, inplace=False):
    _lib.require_gpu_tensor(t, name)
    if t.dtype != torch.float32:
        if inplace:
            raise _lib.VingsError(f"{name} must be float32 (it is updated in place)")
        t = t.float()
    if not t.is_contiguous():
        if inplace:
            raise _lib.VingsError(f"{name} must be contiguous (it is updated in place)")
        t = t.contiguous()
    return t


def _idx(t, dev):
    return t.to(device=dev, dtype=torch.int64).contiguous()


class _Problem:
    """Validated arguments + workspace of one BA call."""

    _ws_cache = {}

    def __init__(self, poses, disps, intr, disps_sens, target, weight, eta, ii, jj, t0, t1,
                 private_ws=False):
        self.poses = _gpu_f32(poses, "poses", inplace=True)
        self.disps = _gpu_f32(disps, "disps", inplace=True)
        dev = self.poses.device
        self.dev = dev
        if self.poses.dim() != 2 or self.poses.shape[1] != 7 or self.disps.dim() != 3 \
                or self.disps.shape[0] != self.poses.shape[0]:
            raise _lib.VingsError("expected poses[B,7] and disps[B,ht,wd]")
        self.B, self.ht, self.wd = self.disps.shape
        self.intr = _gpu_f32(intr, "intrinsics")
        self.sens = None if disps_sens is None else _gpu_f32(disps_sens, "disps_sens")
        self.ii, self.jj = _idx(ii, dev), _idx(jj, dev)
        self.E = int(self.ii.numel())
        self.target = _gpu_f32(target, "target")
        self.weight = _gpu_f32(weight, "weight")
        if self.E and (tuple(self.target.shape) != (self.E, 2, self.ht, self.wd) or
                       self.target.shape != self.weight.shape):
            raise _lib.VingsError("target / weight must be [E,2,ht,wd]")
        self.eta = _gpu_f32(eta, "eta").reshape(-1, self.ht, self.wd)
        self.M = int(self.eta.shape[0])
        self.t0 = int(t0)
        self.t1 = int(t1)
        self.P = self.t1 - self.t0
        lay = _Layout()
        check(_fn("vings_dba_layout_query")(self.ht, self.wd, self.E, self.P, self.M,
                                            ctypes.byref(lay)), "vings_dba_layout_query")
        self.lay = lay
        key = (dev.index, int(lay.total_bytes))
        # ba() is done with its workspace when it returns, so calls of one size share a cached
        # one; a BACore keeps its linearised system there between init / hessian / retract and
        # owns its workspace like the reference's BACore owns its state
        ws = None if private_ws else _Problem._ws_cache.get(key)
        if private_ws:
            ws = torch.empty(int(lay.total_bytes) + 256, dtype=torch.uint8, device=dev)
        elif ws is None:
            ws = torch.empty(int(lay.total_bytes) + 256, dtype=torch.uint8, device=dev)
            if len(_Problem._ws_cache) > 8:
                _Problem._ws_cache.clear()
            _Problem._ws_cache[key] = ws
        self.ws = ws
        self.ws_ptr = (ws.data_ptr() + 255) & ~255
        self.ws_bytes = int(lay.total_bytes)

    def dims(self):
        return (self.B, self.ht, self.wd, self.E, self.M, self.t0, self.t1)

    def view(self, off, dtype, shape):
        base = self.ws_ptr - self.ws.data_ptr() + int(off)
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        return self.ws[base:base + nbytes].view(dtype).view(*shape)


def ba(poses, disps, intrinsics, disps_sens, target, weight, eta, ii, jj, t0, t1, itrs, lm, ep,
       motion_only):
    """Dense bundle adjustment; ``poses`` and ``disps`` are updated in place."""
    pr = _Problem(poses, disps, intrinsics, disps_sens, target, weight, eta, ii, jj, t0, t1)
    check(_fn("vings_dba_ba")(
        ptr(pr.poses), ptr(pr.disps), ptr(pr.intr), ptr(pr.sens), ptr(pr.target), ptr(pr.weight),
        ptr(pr.eta), ptr(pr.ii), ptr(pr.jj), *pr.dims(), int(itrs), float(lm), float(ep),
        1 if motion_only else 0, pr.ws_ptr, pr.ws_bytes, stream_ptr(pr.dev)), "vings_dba_ba")
    global _LAST
    _LAST = pr
    return []


_LAST = None


def last_update():
    """Pose update dx[P,6] (float64) and solver status of the most recent ``ba`` call
    (diagnostics / tests; synchronises)."""
    pr = _LAST
    dx = pr.view(pr.lay.off_dx, torch.float64, (pr.P, 6)).cpu()
    st = pr.view(pr.lay.off_status, torch.int32, (4,)).cpu()
    return dx, {"depth_variables": int(st[0]), "cholesky_failed": bool(st[1]),
                "bad_index": bool(st[2])}


class BACore:
    """Split form of ``ba`` for fusing the visual system with other factors."""

    def __init__(self):
        self.pr = None

    def init(self, poses, disps, intrinsics, disps_sens, target, weight, eta, ii, jj, t0, t1,
             itrs, lm, ep, motion_only):
        self.pr = _Problem(poses, disps, intrinsics, disps_sens, target, weight, eta, ii, jj, t0, t1,
                           private_ws=True)
        self.motion_only = bool(motion_only)
        self._linearize()

    def _linearize(self):
        pr = self.pr
        check(_fn("vings_dba_linearize")(
            ptr(pr.poses), ptr(pr.disps), ptr(pr.intr), ptr(pr.sens), ptr(pr.target),
            ptr(pr.weight), ptr(pr.eta), ptr(pr.ii), ptr(pr.jj), *pr.dims(),
            1 if self.motion_only else 0, pr.ws_ptr, pr.ws_bytes, stream_ptr(pr.dev)),
            "vings_dba_linearize")
        self._fresh = True

    def hessian(self, H, v):
        """Fill the caller's CPU float64 H[6P,6P], v[6P] with the reduced camera system."""
        pr = self.pr
        if not self._fresh:     # poses / depths moved since the last linearisation
            self._linearize()
        n = 6 * pr.P
        Hd = torch.empty((n, n), dtype=torch.float64, device=pr.dev)
        vd = torch.empty(n, dtype=torch.float64, device=pr.dev)
        check(_fn("vings_dba_reduced_system")(*pr.dims(), pr.ws_ptr, pr.ws_bytes, ptr(Hd), ptr(vd),
                                              stream_ptr(pr.dev)), "vings_dba_reduced_system")
        H.copy_(Hd.to(H.device, H.dtype))
        v.copy_(vd.to(v.device, v.dtype))

    def hessian_expand(self, H, v):
        """hessian() plus (Q[M,HW], E[(P+E),6,HW]) for depth covariances."""
        self.hessian(H, v)
        pr = self.pr
        HW = pr.ht * pr.wd
        Q = torch.empty((pr.M, HW), dtype=torch.float32, device=pr.dev)
        E = torch.empty((pr.P + pr.E, 6, HW), dtype=torch.float32, device=pr.dev)
        check(_fn("vings_dba_export")(*pr.dims(), pr.ws_ptr, pr.ws_bytes, ptr(Q), ptr(E),
                                      stream_ptr(pr.dev)), "vings_dba_export")
        return Q, E

    def retract(self, dx):
        pr = self.pr
        dxd = dx.to(device=pr.dev, dtype=torch.float64).contiguous()
        if dxd.numel() != 6 * pr.P:
            raise _lib.VingsError("dx must hold 6*(t1-t0) entries")
        check(_fn("vings_dba_retract")(ptr(pr.poses), ptr(pr.disps), ptr(dxd), *pr.dims(),
                                       1 if self.motion_only else 0, pr.ws_ptr, pr.ws_bytes,
                                       stream_ptr(pr.dev)), "vings_dba_retract")
        self._fresh = False
        return dxd


def frame_distance(poses, disps, intrinsics, ii, jj, beta):
    poses = _gpu_f32(poses, "poses")
    disps = _gpu_f32(disps, "disps")
    dev = poses.device
    ii, jj = _idx(ii, dev), _idx(jj, dev)
    E = int(ii.numel())
    out = torch.empty(E, dtype=torch.float32, device=dev)
    B, ht, wd = disps.shape
    if poses.shape[0] < B:      # the reference passes poses[:counter] with the full disps
        B = poses.shape[0]
    check(_fn("vings_frame_distance")(ptr(poses), ptr(disps), ptr(_gpu_f32(intrinsics, "intr")),
                                      ptr(ii), ptr(jj), B, ht, wd, E, float(beta), ptr(out),
                                      stream_ptr(dev)), "vings_frame_distance")
    return out


def _corr_args(volume, coords, radius):
    _lib.require_gpu_tensor(volume, "volume")
    if volume.dtype not in (torch.float32, torch.float16):
        volume = volume.float()
    volume = volume.contiguous()
    coords = _gpu_f32(coords, "coords")
    E, h1, w1, h2, w2 = volume.shape
    if tuple(coords.shape) != (E, 2, h1, w1):
        raise _lib.VingsError("coords must be [E,2,h1,w1]")
    return volume, coords, (E, h1, w1, h2, w2, int(radius), 1 if volume.dtype == torch.float16 else 0)


def corr_index_forward(volume, coords, radius):
    """droid_backends.corr_index_forward (modules/corr.py:12): [corr] with corr[E,2r+1,2r+1,h1,w1].
    Channel convention: corr[e,i,j,y,x] samples volume[e,y,x] at (coords.x + i - r,
    coords.y + j - r): the X offset is the slow index (the layout of the public DROID-SLAM
    kernels, which the update operator's pretrained weights expect; restated from there, the
    reference tree holds nothing that pins it).  altcorr_forward uses the same order."""
    volume, coords, dims = _corr_args(volume, coords, radius)
    E, h1, w1, _, _, r, _ = dims
    rd = 2 * r + 1
    corr = torch.empty((E, rd, rd, h1, w1), dtype=volume.dtype, device=volume.device)
    check(_fn("vings_corr_index_forward")(ptr(volume), ptr(coords), ptr(corr), *dims,
                                          stream_ptr(volume.device)), "vings_corr_index_forward")
    return [corr]


def corr_index_backward(volume, coords, corr_grad, radius):
    volume, coords, dims = _corr_args(volume, coords, radius)
    corr_grad = corr_grad.to(volume.dtype).contiguous()
    grad = torch.empty_like(volume)
    check(_fn("vings_corr_index_backward")(ptr(coords), ptr(corr_grad), ptr(grad), *dims,
                                           stream_ptr(volume.device)), "vings_corr_index_backward")
    return [grad]


def _altcorr_args(fmap1, fmap2, coords, radius):
    fmap1 = _gpu_f32(fmap1, "fmap1")
    fmap2 = _gpu_f32(fmap2, "fmap2")
    coords = _gpu_f32(coords, "coords")
    if fmap1.dim() != 4 or fmap2.dim() != 4 or coords.dim() != 5:
        raise _lib.VingsError("expected fmap1[B,H,W,C], fmap2[B,H2,W2,C], coords[B,S,H,W,2]")
    B, H, W, C = fmap1.shape
    B2, H2, W2, C2 = fmap2.shape
    if B2 != B or C2 != C or tuple(coords.shape[0:1] + coords.shape[2:]) != (B, H, W, 2):
        raise _lib.VingsError("fmap1 / fmap2 / coords shapes do not belong together")
    return fmap1, fmap2, coords, (B, int(coords.shape[1]), H, W, H2, W2, C, int(radius))


def altcorr_forward(fmap1, fmap2, coords, radius):
    """Correlation window computed on the fly (modules/corr.py:79): fmap1[B,H,W,C],
    fmap2[B,H2,W2,C], coords[B,S,H,W,2] -> 1-tuple (corr[B,S,(2r+1)^2,H,W],)."""
    fmap1, fmap2, coords, dims = _altcorr_args(fmap1, fmap2, coords, radius)
    B, S, H, W, _, _, _, r = dims
    corr = torch.empty((B, S, (2 * r + 1) ** 2, H, W), dtype=torch.float32, device=fmap1.device)
    check(_fn("vings_altcorr_forward")(ptr(fmap1), ptr(fmap2), ptr(coords), ptr(corr), *dims,
                                       stream_ptr(fmap1.device)), "vings_altcorr_forward")
    return [corr]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    """3-tuple (fmap1_grad, fmap2_grad, coords_grad) as unpacked at modules/corr.py:86-87;
    coords_grad is zero (the lookup is not differentiated with respect to its position, and the
    tracker runs under no_grad, dbaf.py:61)."""
    fmap1, fmap2, coords, dims = _altcorr_args(fmap1, fmap2, coords, radius)
    B, S, H, W, _, _, _, r = dims
    corr_grad = _gpu_f32(corr_grad, "corr_grad")
    if tuple(corr_grad.shape) != (B, S, (2 * r + 1) ** 2, H, W):
        raise _lib.VingsError("corr_grad must be [B,S,(2r+1)^2,H,W]")
    g1, g2 = torch.empty_like(fmap1), torch.empty_like(fmap2)
    check(_fn("vings_altcorr_backward")(ptr(fmap1), ptr(fmap2), ptr(coords), ptr(corr_grad), ptr(g1),
                                        ptr(g2), *dims, stream_ptr(fmap1.device)),
          "vings_altcorr_backward")
    return [g1, g2, torch.zeros_like(coords)]


# ---------------------------------------------------------------------------
# benchmark helper
# ---------------------------------------------------------------------------
class BenchProblem:
    """A fixed synthetic window re-solved from the same start every step."""

    def __init__(self, dev, ht, wd, n_frames=9, n_edges=48, itrs=2):
        from . import synth
        p = synth.make_dba_problem(n_frames=n_frames, ht=ht, wd=wd, n_edges=n_edges)
        f = lambda a: torch.from_numpy(a).float().to(dev).contiguous()
        self.np = p
        self.poses0, self.disps0 = f(p["poses"]), f(p["disps"])
        self.poses, self.disps = self.poses0.clone(), self.disps0.clone()
        self.intr, self.sens = f(p["intr"]), f(p["disps_sens"])
        self.target, self.weight, self.eta = f(p["target"]), f(p["weight"]), f(p["eta"])
        self.ii = torch.from_numpy(p["ii"]).to(dev)
        self.jj = torch.from_numpy(p["jj"]).to(dev)
        self.t0, self.t1, self.itrs = p["t0"], p["t1"], itrs
        self.ht, self.wd, self.E = ht, wd, len(p["ii"])

    def run(self):
        self.poses.copy_(self.poses0)
        self.disps.copy_(self.disps0)
        ba(self.poses, self.disps, self.intr, self.sens, self.target, self.weight, self.eta,
           self.ii, self.jj, self.t0, self.t1, self.itrs, 1e-4, 0.1, False)
        self.disps.clamp_(min=0.001)

    def describe(self):
        import numpy as np
        depth_maps = len(np.unique(np.concatenate((np.arange(self.t0, self.t1), self.np["ii"]))))
        return {"grid": f"{self.ht}x{self.wd}", "edges": self.E, "free_poses": self.t1 - self.t0,
                "depth_maps": int(depth_maps), "itrs": self.itrs, "lm": 1e-4, "ep": 0.1}


def make_bench_problem(dev, ht, wd):
    return BenchProblem(dev, ht, wd)
