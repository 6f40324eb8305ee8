"""CPU restatement (numpy) of the tracker -> mapper packaging steps.

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the shipped package.

PARITY UNPINNED.  The reference holds no tests or fixtures for these functions, and their
modules cannot be imported here (depth_video.py needs lietorch, droid_backends and gtsam;
droid_net.py needs lietorch and torch_scatter; middleware_utils.py needs the whole tracker).
Each function below is restated from the reference's text:
  compute_covariances   /root/reference/scripts/frontend/depth_video.py:676-726
  cvx_upsample          /root/reference/scripts/frontend/droid_net.py:17-38
  judge_and_package_v3  /root/reference/scripts/vings_utils/middleware_utils.py:224-238
tests/test_oracle_tracker.py checks the one non-obvious step (the broadcasting assignment of
the per-pose blocks at depth_video.py:714) against PyTorch's own indexing rules.
"""
import numpy as np


def assemble_couplings(E, ii, jj, t0, t1):
    """E_sum[P,K,6,HW] of depth_video.py:703-717.  E = [P per-pose blocks | one per edge]."""
    ii = np.asarray(ii); jj = np.asarray(jj)
    P = t1 - t0
    Ei, Ejz = E[:P], E[P:P + len(ii)]
    assert Ejz.shape[0] == len(ii)
    kx = np.unique(ii)
    K = len(kx)
    mn = min(ii.min(), jj.min())
    HW = E.shape[-1]
    full = np.zeros((K, K, 6, HW), E.dtype)          # [pose frame - mn, depth frame - mn]
    for e in range(len(ii)):
        full[jj[e] - mn, ii[e] - mn] = Ejz[e]        # the last edge of a pair wins
    assert 0 <= t0 - mn and t1 - mn <= K
    out = full[t0 - mn:t1 - mn].copy()               # free poses only: [P,K,6,HW]
    # the reference's assignment pairs range(P) on the pose axis with a SLICE on the depth
    # axis, so Ei[s] lands in column t0-mn+s of EVERY pose row
    for p in range(P):
        for s in range(P):
            out[p, t0 - mn + s] = Ei[s]
    return out, kx


def compute_covariances(H, Q, E, ii, jj, t0, t1, disps, depths_cov):
    """Returns the updated copy of depths_cov ([B,ht,wd]); float32 arithmetic like the reference
    (H is cast from float64 before the factorisation)."""
    from scipy.linalg import solve_triangular
    P = t1 - t0
    L = np.linalg.cholesky(np.asarray(H, np.float32))
    L_inv = solve_triangular(L, np.eye(6 * P, dtype=np.float32), lower=True).astype(np.float32)
    E_sum, kx = assemble_couplings(np.asarray(E, np.float32), ii, jj, t0, t1)
    K, HW = len(kx), E.shape[-1]
    A = E_sum.transpose(0, 2, 1, 3).reshape(6 * P, K * HW)            # rows (p,d), columns (k,pix)
    Qv = np.asarray(Q, np.float32).reshape(K * HW, 1)
    F = (Qv * A.T).astype(np.float64) @ L_inv.astype(np.float64)      # wide accumulation
    z = Qv[:, 0].astype(np.float64) + (F ** 2).sum(-1)
    out = np.array(depths_cov, np.float32, copy=True)
    d = np.asarray(disps, np.float64)[kx].reshape(K, HW)
    out[kx] = (z.reshape(K, HW) / (d ** 4 + 1e-6)).reshape((K,) + out.shape[1:]).astype(np.float32)
    return out


def cvx_upsample(data, mask):
    """data [B,ht,wd,dim], mask [B,576,ht,wd] -> [B,8ht,8wd,dim]; float64."""
    data = np.asarray(data, np.float64)
    B, ht, wd, dim = data.shape
    m = np.asarray(mask, np.float64).reshape(B, 9, 8, 8, ht, wd)
    m = np.exp(m - m.max(axis=1, keepdims=True))
    m /= m.sum(axis=1, keepdims=True)
    pad = np.zeros((B, ht + 2, wd + 2, dim))
    pad[:, 1:-1, 1:-1] = data
    out = np.zeros((B, ht, 8, wd, 8, dim))
    for n in range(9):
        dy, dx = n // 3, n % 3
        nb = pad[:, dy:dy + ht, dx:dx + wd]                          # [B,ht,wd,dim]
        w = m[:, n].transpose(0, 3, 1, 4, 2)                          # [B,ht,8(i),wd,8(j)]
        out += w[..., None] * nb[:, :, None, :, None, :]
    return out.reshape(B, 8 * ht, 8 * wd, dim)


def package_keyframes(images, disps_up, depths_cov_up, ids, max_depth, cov_times):
    """float32 like the reference.  Returns rgbs [n,H,W,3], depths [n,H,W,1], cov [n,H,W,1], median [n]."""
    ids = np.asarray(ids)
    # on the device PyTorch divides by a scalar as a multiplication by its float32 reciprocal
    rgbs = np.asarray(images)[ids].transpose(0, 2, 3, 1).astype(np.float32) * (np.float32(1.0) / np.float32(255.0))
    depths = (np.float32(1.0) / (np.asarray(disps_up, np.float32)[ids] + np.float32(1e-6)))[..., None]
    cov = np.array(np.asarray(depths_cov_up, np.float32)[ids][..., None], copy=True)
    n = len(ids)
    med = np.median(cov.reshape(n, -1), axis=1).astype(np.float32)
    zero = (depths > np.float32(max_depth)) | (cov > np.float32(cov_times) * med[:, None, None, None])
    depths[zero] = 0.0
    cov[depths == 0] = cov[depths > 0].max()
    rgbs[depths[..., 0] == 0] = 0.0
    return rgbs, depths, cov, med
