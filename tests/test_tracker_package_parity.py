"""GPU parity of the tracker -> mapper packaging kernels (csrc/tracker_package.hip) against
the CPU oracle (oracle/tracker_oracle.py).  Selections, masks and copies are bit-exact;
floating-point results within the tolerance written at each check."""
import numpy as np
import pytest
import torch

from oracle import tracker_oracle as T
from vings_mono_amd import _lib, tracker_package as TP

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(lo, hi, seed=0):
    rng = np.random.default_rng(seed)
    ii, jj = [], []
    for i in range(lo, hi):
        for j in range(lo, hi):
            if 1 <= abs(i - j) <= 3:
                ii.append(i); jj.append(j)
    order = rng.permutation(len(ii))
    return np.array(ii)[order], np.array(jj)[order]


@pytest.mark.parametrize("shape", [(4, 6, 3, 9), (11, 19, 2, 12)])   # ht, wd, t0, t1
def test_depth_covariance_matches_oracle(shape):
    """float32 kernel vs wide-accumulation oracle: 1e-4 relative."""
    ht, wd, t0, t1 = shape
    lo, hi = t0 - 1, t1 + 1                       # one fixed frame on either side
    ii, jj = _graph(lo, hi, seed=ht)
    P, M, HW, B = t1 - t0, len(ii), ht * wd, hi + 3
    rng = np.random.default_rng(ht * 7)
    A = rng.normal(size=(6 * P, 6 * P))
    H = A @ A.T + 6 * P * np.eye(6 * P)
    E = rng.normal(size=(P + M, 6, HW)).astype(np.float32)
    kx = np.unique(ii)
    Q = rng.uniform(0.1, 1.0, (len(kx), HW)).astype(np.float32)
    disps = rng.uniform(0.3, 1.0, (B, ht, wd)).astype(np.float32)
    cov0 = np.full((B, ht, wd), 100.0, np.float32)
    want = T.compute_covariances(H, Q, E, ii, jj, t0, t1, disps, cov0)
    cov = torch.from_numpy(cov0).to(DEV)
    ok = TP.compute_covariances(torch.from_numpy(H), torch.from_numpy(Q).to(DEV),
                                torch.from_numpy(E).to(DEV), ht, wd, torch.from_numpy(ii).to(DEV),
                                torch.from_numpy(jj).to(DEV), t0, t1,
                                torch.from_numpy(disps).to(DEV), cov)
    assert ok
    np.testing.assert_allclose(cov.cpu().numpy(), want, rtol=1e-4)
    # no Cholesky factor: nothing is written, as the reference skips silently
    cov2 = torch.from_numpy(cov0).to(DEV)
    assert not TP.compute_covariances(-torch.eye(6 * P), torch.from_numpy(Q).to(DEV),
                                      torch.from_numpy(E).to(DEV), ht, wd, ii, jj, t0, t1,
                                      torch.from_numpy(disps).to(DEV), cov2)
    assert torch.equal(cov2.cpu(), torch.from_numpy(cov0))


@pytest.mark.parametrize("shape", [(5, 7), (43, 77), (16, 64)])
def test_upsample_matches_oracle(shape):
    """1e-5 relative + 1e-6 absolute (float32 softmax vs float64 oracle)."""
    ht, wd = shape
    B, K = 6, 3
    g = torch.Generator().manual_seed(ht)
    disps = torch.rand(B, ht, wd, generator=g)
    cov = torch.rand(B, ht, wd, generator=g) * 5
    mask = torch.randn(1, K, 576, ht, wd, generator=g) * 3
    ix = torch.tensor([4, 0, 2])
    up_d = torch.full((B, 8 * ht, 8 * wd), -1.0, device=DEV)
    up_c = torch.full((B, 8 * ht, 8 * wd), -1.0, device=DEV)
    TP.upsample_maps(ix, mask.to(DEV), [disps.to(DEV), cov.to(DEV)], [up_d, up_c])
    want_d = T.cvx_upsample(disps[ix][..., None].numpy(), mask[0].numpy())[..., 0]
    want_c = T.cvx_upsample(cov[ix][..., None].numpy(), mask[0].numpy())[..., 0]
    np.testing.assert_allclose(up_d[ix].cpu().numpy(), want_d, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(up_c[ix].cpu().numpy(), want_c, rtol=1e-5, atol=1e-6)
    rest = [f for f in range(B) if f not in ix.tolist()]
    assert float(up_d[rest].max()) == -1.0 == float(up_d[rest].min())     # other frames untouched
    # the generic entry: channel-last data with two channels
    both = torch.stack([disps[ix], cov[ix]], dim=-1).to(DEV)
    got = TP.cvx_upsample(both, mask.to(DEV))
    np.testing.assert_allclose(got[..., 1].cpu().numpy(), want_c, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(24, 40, 3), (57, 91, 2), (344, 616, 4)])   # odd and even pixel counts
def test_package_is_bit_exact(shape):
    H, W, n = shape
    B = 6
    rng = np.random.default_rng(H)
    images = rng.integers(0, 256, (B, 3, H, W), dtype=np.uint8)
    disps = rng.uniform(0.01, 1.0, (B, H, W)).astype(np.float32)
    disps[:, :2] = 0.0                                           # no disparity: depth 1e6, dropped
    cov = (rng.uniform(0.0, 2.0, (B, H, W)) ** 3).astype(np.float32)
    cov[2, 5:9] = cov[2, 4, 7]                                   # repeated values around the median
    ids = [5, 2, 0, 3][:n]
    want = T.package_keyframes(images, disps, cov, ids, max_depth=30.0, cov_times=2.0)
    got = TP.package_keyframes(torch.from_numpy(images).to(DEV), torch.from_numpy(disps).to(DEV),
                               torch.from_numpy(cov).to(DEV), ids, 30.0, 2.0)
    np.testing.assert_array_equal(got["cov_median"].cpu().numpy(), want[3])
    np.testing.assert_array_equal(got["depths"].cpu().numpy(), want[1])
    np.testing.assert_array_equal(got["images"].cpu().numpy(), want[0])
    np.testing.assert_array_equal(got["depths_cov"].cpu().numpy(), want[2])
    assert got["pixel_mask"].all()


def test_package_median_with_negative_and_constant_maps():
    B, H, W = 3, 16, 16
    cov = np.zeros((B, H, W), np.float32)
    cov[0] = 7.5                                                  # constant map
    cov[1] = np.linspace(-3, 3, H * W, dtype=np.float32).reshape(H, W)   # both signs, even count
    cov[2] = np.random.default_rng(1).normal(size=(H, W)).astype(np.float32)
    images = np.zeros((B, 3, H, W), np.uint8)
    disps = np.full((B, H, W), 0.5, np.float32)
    got = TP.package_keyframes(torch.from_numpy(images).to(DEV), torch.from_numpy(disps).to(DEV),
                               torch.from_numpy(cov).to(DEV), [0, 1, 2], 100.0, 1e9)
    np.testing.assert_array_equal(got["cov_median"].cpu().numpy(),
                                  np.median(cov.reshape(B, -1), axis=1).astype(np.float32))


def test_package_at_full_resolution_properties():
    """1080p, 8 frames: the median against a device sort, the masks against their definition."""
    n, H, W = 8, 1080, 1920
    g = torch.Generator(device=DEV).manual_seed(4)
    images = torch.randint(0, 256, (n, 3, H, W), device=DEV, dtype=torch.uint8, generator=g)
    disps = torch.rand((n, H, W), device=DEV, generator=g) + 0.01
    cov = torch.rand((n, H, W), device=DEV, generator=g) ** 4
    got = TP.package_keyframes(images, disps, cov, list(range(n)), 40.0, 3.0)
    srt = cov.reshape(n, -1).sort(dim=1).values
    m = H * W
    med = (srt[:, m // 2 - 1] + srt[:, m // 2]) / 2
    assert torch.equal(got["cov_median"], med)
    depth = 1.0 / (disps + 1e-6)
    zero = (depth > 40.0) | (cov > 3.0 * med[:, None, None])
    assert torch.equal(got["depths"][..., 0] == 0, zero)
    assert torch.equal(got["depths"][..., 0][~zero], depth[~zero])
    top = cov[~zero].max()
    assert torch.equal(got["depths_cov"][..., 0], torch.where(zero, top, cov))
    assert float(got["images"][zero].abs().max()) == 0.0
    want_rgb = images.permute(0, 2, 3, 1).float() / 255.0
    assert torch.equal(got["images"][~zero], want_rgb[~zero])
