"""CPU checks of the tracker-packaging oracle (oracle/tracker_oracle.py) and of the host
logic that mirrors it in the package.  The reference has no fixtures for these steps (parity
unpinned); the checks below tie the restatement to PyTorch's own operator semantics."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tracker_oracle as T
from vings_mono_amd import tracker_package as TP


def _graph(t0=2, t1=6, lo=1, hi=7, seed=0):
    """Edges among contiguous frames lo..hi-1, |i-j| in {1,2}, shuffled."""
    rng = np.random.default_rng(seed)
    ii, jj = [], []
    for i in range(lo, hi):
        for j in range(lo, hi):
            if 1 <= abs(i - j) <= 2:
                ii.append(i); jj.append(j)
    order = rng.permutation(len(ii))
    return np.array(ii)[order], np.array(jj)[order], t0, t1


def test_coupling_assembly_follows_torch_indexing_rules():
    ii, jj, t0, t1 = _graph()
    P, M, HW = t1 - t0, len(ii), 5
    rng = np.random.default_rng(1)
    E = rng.normal(size=(P + M, 6, HW)).astype(np.float32)
    got, kx = T.assemble_couplings(E, ii, jj, t0, t1)
    # the same steps with torch's advanced indexing: scatter the edge blocks into a K x K
    # table, keep the free-pose rows, then assign the per-pose blocks through
    # (list on axis 0, slice on axis 1), which broadcasts over axis 0
    K, mn = len(kx), min(ii.min(), jj.min())
    Et = torch.from_numpy(E)
    tab = torch.zeros(K, K, 6, HW)
    tab[torch.from_numpy(jj - mn), torch.from_numpy(ii - mn)] = Et[P:]
    tab = tab[t0 - mn:t1 - mn].reshape(P, K, 6, HW)
    tab[list(range(P)), t0 - mn:t1 - mn] = Et[:P]
    np.testing.assert_array_equal(got, tab.numpy())
    # and it is NOT block diagonal: pose 0 sees the block of pose 1 in pose 1's column
    np.testing.assert_array_equal(got[0, t0 - mn + 1], E[1])


def test_coupling_table_names_the_same_blocks():
    ii, jj, t0, t1 = _graph(seed=3)
    P, M, HW = t1 - t0, len(ii), 3
    E = np.random.default_rng(2).normal(size=(P + M, 6, HW)).astype(np.float32)
    want, kx = T.assemble_couplings(E, ii, jj, t0, t1)
    table, kx2 = TP.coupling_table(ii, jj, t0, t1)
    np.testing.assert_array_equal(kx, kx2)
    for k in range(len(kx)):
        for p in range(P):
            blk = table[k, p]
            np.testing.assert_array_equal(want[p, k], E[blk] if blk >= 0 else np.zeros((6, HW)))
    with pytest.raises(IndexError):
        TP.coupling_table(np.array([1, 5]), np.array([5, 1]), 2, 6)      # frames with a gap


def test_covariance_oracle_equals_dense_quadratic_form():
    ii, jj, t0, t1 = _graph()
    P, M, HW, B = t1 - t0, len(ii), 4, 8
    rng = np.random.default_rng(5)
    A = rng.normal(size=(6 * P, 6 * P))
    H = A @ A.T + 6 * P * np.eye(6 * P)
    E = rng.normal(size=(P + M, 6, HW)).astype(np.float32)
    kx = np.unique(ii)
    Q = rng.uniform(0.1, 1.0, (len(kx), HW)).astype(np.float32)
    disps = rng.uniform(0.3, 1.0, (B, 2, 2)).astype(np.float32)
    cov0 = np.full((B, 2, 2), 100.0, np.float32)
    got = T.compute_covariances(H, Q, E, ii, jj, t0, t1, disps, cov0)
    E_sum, _ = T.assemble_couplings(E, ii, jj, t0, t1)
    Linv = np.linalg.inv(np.linalg.cholesky(H))
    G = Linv @ Linv.T                                  # |e^T L^-1|^2 = e^T (L^-1 L^-T) e
    for k, frame in enumerate(kx):
        for pix in range(HW):
            e = Q[k, pix] * E_sum[:, k, :, pix].reshape(-1).astype(np.float64)
            z = Q[k, pix] + e @ G @ e
            want = z / (float(disps[frame].reshape(-1)[pix]) ** 4 + 1e-6)
            assert abs(got[frame].reshape(-1)[pix] - want) <= 1e-4 * abs(want)
    untouched = [f for f in range(B) if f not in kx]
    assert (got[untouched] == 100.0).all()


def test_cvx_upsample_oracle_equals_unfold_formulation():
    g = torch.Generator().manual_seed(3)
    B, ht, wd, dim = 2, 5, 7, 2
    data = torch.rand(B, ht, wd, dim, generator=g, dtype=torch.float64)
    mask = torch.randn(B, 576, ht, wd, generator=g, dtype=torch.float64) * 2
    w = torch.softmax(mask.view(B, 1, 9, 8, 8, ht, wd), dim=2)
    nb = F.unfold(data.permute(0, 3, 1, 2), [3, 3], padding=1).view(B, dim, 9, 1, 1, ht, wd)
    up = (w * nb).sum(dim=2).permute(0, 4, 2, 5, 3, 1).reshape(B, 8 * ht, 8 * wd, dim)
    np.testing.assert_allclose(T.cvx_upsample(data.numpy(), mask.numpy()), up.numpy(), atol=1e-12)


def test_package_oracle_masks_and_fills():
    rng = np.random.default_rng(8)
    B, H, W = 4, 6, 8
    images = rng.integers(0, 256, (B, 3, H, W), dtype=np.uint8)
    disps = rng.uniform(0.01, 1.0, (B, H, W)).astype(np.float32)
    cov = rng.uniform(0.0, 2.0, (B, H, W)).astype(np.float32)
    rgbs, depths, c, med = T.package_keyframes(images, disps, cov, [1, 3], max_depth=20.0, cov_times=1.5)
    assert rgbs.shape == (2, H, W, 3) and depths.shape == (2, H, W, 1)
    gone = depths[..., 0] == 0
    assert gone.any() and (~gone).any()
    assert (rgbs[gone] == 0).all() and (c[..., 0][gone] == c[..., 0][~gone].max()).all()
    np.testing.assert_array_equal(med, np.median(cov[[1, 3]].reshape(2, -1), axis=1))
    far = (1.0 / (disps[[1, 3]] + 1e-6)) > 20.0
    assert gone[far].all()
