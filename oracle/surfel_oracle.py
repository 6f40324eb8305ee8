"""ctypes front end of the C surfel-rasteriser oracle (oracle/surfel_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The shipped package never imports this module.
Parity status and the reference call sites it restates are in the header of
surfel_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
N_AUX = 9


def build(force=False):
    """Compile both precisions of the oracle with gcc (seconds)."""
    need = force or not all(
        os.path.exists(os.path.join(_BUILD, f"libsurfel_oracle_{p}.so")) for p in ("f32", "f64"))
    if not need:
        src = os.path.getmtime(os.path.join(_HERE, "surfel_oracle.c"))
        need = any(os.path.getmtime(os.path.join(_BUILD, f"libsurfel_oracle_{p}.so")) < src
                   for p in ("f32", "f64"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s"])


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class SurfelOracle:
    """Numpy-in / numpy-out wrapper. dtype float32 (parity) or float64 (gradient checks)."""

    def __init__(self, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        tag = "f32" if self.dtype == np.float32 else "f64"
        path = os.path.join(_BUILD, f"libsurfel_oracle_{tag}.so")
        if not os.path.exists(path):
            build()
        self.lib = ctypes.CDLL(path)
        self.lib.surfel_bin_count.restype = ctypes.c_int64
        assert self.lib.oracle_real_bytes() == self.dtype.itemsize
        self.creal = ctypes.c_float if tag == "f32" else ctypes.c_double

    def _a(self, x, shape=None):
        if x is None:
            return None
        a = np.ascontiguousarray(np.asarray(x, dtype=self.dtype))
        if shape is not None:
            a = a.reshape(shape)
        return a

    # -- stages ---------------------------------------------------------
    def project(self, means3D, scales, rots, opac, view, proj, H, W, scale_mod=1.0):
        N = means3D.shape[0]
        f = self.dtype
        out = dict(
            xy=np.zeros((N, 2), f), depth=np.zeros(N, f), transmat=np.zeros((N, 9), f),
            normal_opacity=np.zeros((N, 4), f), radii=np.zeros(N, np.int32),
            rect=np.zeros((N, 4), np.int32), tiles_touched=np.zeros(N, np.uint32))
        rc = self.lib.surfel_project(
            N, H, W, self.creal(scale_mod), _ptr(means3D), _ptr(scales), _ptr(rots), _ptr(opac),
            _ptr(view), _ptr(proj), _ptr(out["xy"]), _ptr(out["depth"]), _ptr(out["transmat"]),
            _ptr(out["normal_opacity"]), _ptr(out["radii"]), _ptr(out["rect"]),
            _ptr(out["tiles_touched"]))
        assert rc == 0
        return out

    def bin(self, pr, H, W):
        N = pr["depth"].shape[0]
        D = int(self.lib.surfel_bin_count(N, _ptr(pr["tiles_touched"])))
        T = ((W + 15) // 16) * ((H + 15) // 16)
        keys = np.zeros(max(D, 1), np.uint64)
        vals = np.zeros(max(D, 1), np.uint32)
        ranges = np.zeros((T, 2), np.uint32)
        rc = self.lib.surfel_bin(N, H, W, _ptr(pr["depth"]), _ptr(pr["rect"]),
                                 _ptr(pr["tiles_touched"]), ctypes.c_int64(D), _ptr(keys),
                                 _ptr(vals), _ptr(ranges))
        assert rc == 0
        return dict(D=D, keys=keys[:D], vals=vals[:D], ranges=ranges)

    # -- full passes ------------------------------------------------------
    def forward(self, means3D, scales, rots, opac, colors, view, proj, bg, H, W,
                scale_mod=1.0, pixel_mask=None, flow=None):
        means3D = self._a(means3D, (-1, 3)); scales = self._a(scales, (-1, 2))
        rots = self._a(rots, (-1, 4)); opac = self._a(opac, (-1,)); colors = self._a(colors, (-1, 3))
        view = self._a(view, (16,)); proj = self._a(proj, (16,)); bg = self._a(bg, (3,))
        flow = self._a(flow, (-1, 2)) if flow is not None else None
        if pixel_mask is not None:
            pixel_mask = np.ascontiguousarray(np.asarray(pixel_mask).reshape(-1).astype(np.uint8))
        pr = self.project(means3D, scales, rots, opac, view, proj, H, W, scale_mod)
        bn = self.bin(pr, H, W)
        f = self.dtype
        HW = H * W
        color = np.zeros((3, H, W), f); allmap = np.zeros((N_AUX, H, W), f)
        final_T = np.zeros((3, H, W), f); n_contrib = np.zeros((2, H, W), np.uint32)
        vals = bn["vals"] if bn["D"] > 0 else np.zeros(1, np.uint32)
        rc = self.lib.surfel_blend_forward(
            H, W, _ptr(bn["ranges"]), _ptr(vals), _ptr(pr["xy"]), _ptr(pr["transmat"]),
            _ptr(pr["normal_opacity"]), _ptr(colors), _ptr(flow), _ptr(bg), _ptr(pixel_mask),
            _ptr(color), _ptr(allmap), _ptr(final_T), _ptr(n_contrib))
        assert rc == 0
        st = dict(H=H, W=W, scale_mod=scale_mod, means3D=means3D, scales=scales, rots=rots,
                  opac=opac, colors=colors, view=view, proj=proj, bg=bg, flow=flow,
                  pixel_mask=pixel_mask, color=color, allmap=allmap, final_T=final_T,
                  n_contrib=n_contrib, radii=pr["radii"], **{k: pr[k] for k in
                  ("xy", "depth", "transmat", "normal_opacity", "rect", "tiles_touched")},
                  D=bn["D"], keys=bn["keys"], vals=bn["vals"], ranges=bn["ranges"])
        return st

    def backward(self, st, dL_dcolor, dL_dallmap):
        f = self.dtype
        N = st["means3D"].shape[0]
        H, W = st["H"], st["W"]
        dL_dcolor = self._a(dL_dcolor, (3, H, W)); dL_dallmap = self._a(dL_dallmap, (N_AUX, H, W))
        g = dict(transmat=np.zeros((N, 9), f), xy=np.zeros((N, 2), f), normal=np.zeros((N, 3), f),
                 opacity=np.zeros(N, f), colors=np.zeros((N, 3), f), scores=np.zeros((N, 2), f),
                 means3D=np.zeros((N, 3), f), scales=np.zeros((N, 2), f), rots=np.zeros((N, 4), f))
        vals = st["vals"] if st["D"] > 0 else np.zeros(1, np.uint32)
        rc = self.lib.surfel_blend_backward(
            H, W, _ptr(st["ranges"]), _ptr(vals), _ptr(st["xy"]), _ptr(st["transmat"]),
            _ptr(st["normal_opacity"]), _ptr(st["colors"]), _ptr(st["flow"]), _ptr(st["bg"]),
            _ptr(st["pixel_mask"]), _ptr(st["final_T"]), _ptr(st["n_contrib"]), _ptr(dL_dcolor),
            _ptr(dL_dallmap), _ptr(g["transmat"]), _ptr(g["xy"]), _ptr(g["normal"]),
            _ptr(g["opacity"]), _ptr(g["colors"]), _ptr(g["scores"]))
        assert rc == 0
        rc = self.lib.surfel_project_backward(
            N, H, W, self.creal(st["scale_mod"]), _ptr(st["means3D"]), _ptr(st["scales"]),
            _ptr(st["rots"]), _ptr(st["view"]), _ptr(st["proj"]), _ptr(st["radii"]),
            _ptr(g["transmat"]), _ptr(g["xy"]), _ptr(g["normal"]), _ptr(g["means3D"]),
            _ptr(g["scales"]), _ptr(g["rots"]))
        assert rc == 0
        return g

    def sparse_adam(self, param, grad, exp_avg, exp_avg_sq, visible, lr, b1=0.9, b2=0.999,
                    eps=1e-15):
        """In place on param / exp_avg / exp_avg_sq (numpy arrays of self.dtype, [N,M])."""
        N = visible.shape[0]
        M = param.size // N
        vis = np.ascontiguousarray(visible.astype(np.uint8))
        rc = self.lib.sparse_adam(N, M, _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                  _ptr(vis), self.creal(lr), self.creal(b1), self.creal(b2),
                                  self.creal(eps))
        assert rc == 0
