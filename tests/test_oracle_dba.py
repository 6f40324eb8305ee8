"""CPU checks of the dense-BA oracle (no GPU).

The reference has no tests or golden vectors for this path.  Its PyTorch twin of the
linear solve (frontend/geom/chol.py) does import here, and tests/golden/make_dba_golden.py
ran it to produce tests/golden/dba_solve_golden.npz: that pins damping, Schur complement,
Cholesky and back-substitution.  The geometric half (frontend/geom/ba.py, projective_ops.py)
needs lietorch / torch_scatter, stays unpinned and is validated against first principles.
"""
import os

import numpy as np
import pytest

from oracle import dba_oracle as O
from vings_mono_amd import se3np, synth


@pytest.fixture(scope="module")
def prob():
    return synth.make_dba_problem(n_frames=6, ht=12, wd=16, n_edges=20, seed=5)


def test_se3_exp_inverse_and_product():
    rng = np.random.default_rng(0)
    xi = rng.normal(size=(5, 6)) * 0.3
    G = O.se3_exp(xi)
    np.testing.assert_allclose(np.linalg.norm(G[:, 3:], axis=1), 1.0, atol=1e-12)
    I = O.se3_mul(G, O.se3_inv(G))
    np.testing.assert_allclose(I, np.tile([0, 0, 0, 0, 0, 0, 1.0], (5, 1)), atol=1e-12)
    # exp(a xi) exp(b xi) = exp((a+b) xi)
    np.testing.assert_allclose(O.se3_mul(O.se3_exp(0.3 * xi), O.se3_exp(0.7 * xi)), G, atol=1e-12)
    # small-angle branch is continuous
    tiny = np.array([[0.1, -0.2, 0.3, 1e-9, -1e-9, 2e-9]])
    np.testing.assert_allclose(O.se3_exp(tiny)[0, :3], tiny[0, :3], atol=1e-8)


def test_product_helper_agrees_with_oracle(prob):
    c0, v0 = se3np.reproject(prob["poses"], prob["disps"], prob["intr"], prob["ii"], prob["jj"])
    c1, v1 = O.reproject(prob["poses"], prob["disps"], prob["intr"], prob["ii"], prob["jj"])
    np.testing.assert_allclose(c0, c1, atol=1e-9)
    np.testing.assert_array_equal(v0, v1 > 0)


def test_jacobians_match_finite_differences(prob):
    p = prob
    ii, jj = p["ii"], p["jj"]
    coords, valid, (Ji, Jj, Jz) = O.reproject(p["poses"], p["disps"], p["intr"], ii, jj, jacobian=True)
    rng = np.random.default_rng(1)
    h = 1e-6
    for frame in (1, 3):
        xi = rng.normal(size=6)
        pp = p["poses"].copy(); pm = p["poses"].copy()
        pp[frame] = O.se3_mul(O.se3_exp(h * xi), p["poses"][frame])
        pm[frame] = O.se3_mul(O.se3_exp(-h * xi), p["poses"][frame])
        cp, _ = O.reproject(pp, p["disps"], p["intr"], ii, jj)
        cm, _ = O.reproject(pm, p["disps"], p["intr"], ii, jj)
        fd = (cp - cm) / (2 * h)
        an = np.zeros_like(fd)
        for e in range(len(ii)):
            if ii[e] == frame:
                an[e] += Ji[e] @ xi
            if jj[e] == frame:
                an[e] += Jj[e] @ xi
        np.testing.assert_allclose(fd, an, rtol=1e-5, atol=1e-5)
    # inverse depth
    frame = 2
    dd = rng.normal(size=p["disps"].shape[1:]) * 0.1
    dp = p["disps"].copy(); dm = p["disps"].copy()
    dp[frame] += h * dd; dm[frame] -= h * dd
    cp, _ = O.reproject(p["poses"], dp, p["intr"], ii, jj)
    cm, _ = O.reproject(p["poses"], dm, p["intr"], ii, jj)
    fd = (cp - cm) / (2 * h)
    an = np.zeros_like(fd)
    for e in range(len(ii)):
        if ii[e] == frame:
            an[e] = Jz[e] * dd[..., None]
    np.testing.assert_allclose(fd, an, rtol=1e-5, atol=1e-5)


def test_schur_solve_equals_dense_normal_equations(prob):
    p = prob
    s = O.build_system(p["poses"], p["disps"], p["intr"], p["disps_sens"], p["target"],
                       p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"])
    lm, ep = 1e-4, 0.1
    S, rhs, E, Q = O.reduced_system(s, lm, ep)
    dx = O.solve_spd(S, rhs)
    dz = Q * (s["w"].reshape(-1) - E.T @ dx)
    # brute force: assemble the full (poses + depths) normal equations from the raw
    # per-pixel Jacobians and solve them densely
    P, M, HW = s["P"], s["M"], s["HW"]
    n = 6 * P + M * HW
    A = np.zeros((n, n)); b = np.zeros(n)
    coords, valid, (Ji, Jj, Jz) = O.reproject(p["poses"], p["disps"], p["intr"], p["ii"], p["jj"], True)
    tgt = p["target"].transpose(0, 2, 3, 1); wgt = p["weight"].transpose(0, 2, 3, 1)
    kx, kk = O.depth_variables(p["ii"], p["t0"], p["t1"])
    for e in range(len(p["ii"])):
        i, j = p["ii"][e] - p["t0"], p["jj"][e] - p["t0"]
        for pix in range(HW):
            y, x = divmod(pix, p["disps"].shape[2])
            for c in range(2):
                w = 0.001 * valid[e, y, x] * wgt[e, y, x, c]
                if w == 0:
                    continue
                row = np.zeros(n)
                if 0 <= i < P:
                    row[6 * i:6 * i + 6] += Ji[e, y, x, c]
                if 0 <= j < P:
                    row[6 * j:6 * j + 6] += Jj[e, y, x, c]
                row[6 * P + kk[e] * HW + pix] += Jz[e, y, x, c]
                r = tgt[e, y, x, c] - coords[e, y, x, c]
                nz = np.nonzero(row)[0]
                A[np.ix_(nz, nz)] += w * np.outer(row[nz], row[nz])
                b[nz] += w * row[nz] * r
    dpose = np.diag(A)[:6 * P].copy()
    A[np.arange(6 * P), np.arange(6 * P)] += ep + lm * dpose
    A[np.arange(6 * P, n), np.arange(6 * P, n)] += p["eta"].reshape(-1)
    sol = np.linalg.solve(A, b)
    np.testing.assert_allclose(dx, sol[:6 * P], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(dz, sol[6 * P:], rtol=1e-6, atol=1e-9)


def test_zero_residual_is_a_fixed_point():
    p = synth.make_dba_problem(n_frames=5, ht=10, wd=12, n_edges=12, seed=9, noise_px=0.0)
    poses, disps, dx, dz = O.ba(p["poses"], p["disps"], p["intr"], p["disps_sens"], p["target"],
                                p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], itrs=2)
    assert np.abs(dx).max() < 1e-9 and np.abs(dz).max() < 1e-9
    np.testing.assert_allclose(poses, p["poses"], atol=1e-9)
    np.testing.assert_allclose(disps, p["disps"], atol=1e-9)


def test_ba_reduces_reprojection_error():
    p = synth.make_dba_problem(n_frames=6, ht=12, wd=16, n_edges=20, seed=5, noise_px=0.0)
    rng = np.random.default_rng(3)
    noisy = p["poses"].copy()
    noisy[1:6] = O.se3_mul(O.se3_exp(rng.normal(size=(5, 6)) * 0.01), noisy[1:6])

    def cost(poses, disps):
        c, v = O.reproject(poses, disps, p["intr"], p["ii"], p["jj"])
        r = p["target"].transpose(0, 2, 3, 1) - c
        return (0.001 * v[..., None] * p["weight"].transpose(0, 2, 3, 1) * r * r).sum()

    c0 = cost(noisy, p["disps"])
    poses, disps, _, _ = O.ba(noisy, p["disps"], p["intr"], p["disps_sens"], p["target"],
                              p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], itrs=4)
    assert cost(poses, disps) < 0.5 * c0
    # gauge frame untouched
    np.testing.assert_array_equal(poses[0], noisy[0])


def test_motion_only_and_sensor_prior(prob):
    p = prob
    poses, disps, dx, dz = O.ba(p["poses"], p["disps"], p["intr"], p["disps_sens"], p["target"],
                                p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"],
                                itrs=1, motion_only=True)
    assert dz is None
    np.testing.assert_array_equal(disps, p["disps"])
    sens = p["disps"] * 1.1
    _, d2, _, dz2 = O.ba(p["poses"], p["disps"], p["intr"], sens, p["target"], p["weight"],
                         p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], itrs=1)
    _, d1, _, dz1 = O.ba(p["poses"], p["disps"], p["intr"], p["disps_sens"], p["target"],
                         p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], itrs=1)
    # the prior pulls inverse depths toward the (larger) sensor values
    assert (d2 - d1).mean() > 0


def test_frame_distance_properties(prob):
    p = prob
    d = O.frame_distance(p["poses"], p["disps"], p["intr"], p["ii"], p["jj"], 0.3)
    assert d.shape == p["ii"].shape and np.all(d > 0)
    same = O.frame_distance(p["poses"], p["disps"], p["intr"], np.array([2]), np.array([2]), 1.0)
    # self edge = fixed stereo baseline 0.1 m: flow = fx * 0.1 * disp on average
    fx = p["intr"][0]
    np.testing.assert_allclose(same[0], (fx * 0.1 * p["disps"][2]).mean(), rtol=1e-6)
    far = p["poses"].copy(); far[4, 2] -= 1e3   # camera 4 far behind: nothing in front of it
    d2 = O.frame_distance(far, p["disps"], p["intr"], np.array([0]), np.array([4]), 0.3)
    assert d2[0] == 1000.0


def test_corr_index_bilinear_and_adjoint():
    rng = np.random.default_rng(4)
    E, h, w = 2, 5, 6
    vol = rng.normal(size=(E, h, w, h, w))
    # integer coordinates reproduce the volume entries
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    coords = np.stack([np.broadcast_to(xx, (E, h, w)), np.broadcast_to(yy, (E, h, w))], 1).astype(float)
    out = O.corr_index_forward(vol, coords, 1)
    assert out.shape == (E, 3, 3, h, w)
    e = 1
    np.testing.assert_allclose(out[e, 1, 1], vol[e, yy, xx, yy, xx])
    # X offset is the slow window index: element (2, 1) looks one pixel to the RIGHT
    np.testing.assert_allclose(out[e, 2, 1, :, :-1], vol[e, yy, xx, yy, np.minimum(xx + 1, w - 1)][:, :-1])
    assert np.all(out[e, :, 0, 0, :] == 0)   # window entries above the map (Y offset -1 in row 0) are zero
    # adjoint identity <A v, g> = <v, A^T g> at random sub-pixel coordinates
    coords = coords + rng.uniform(-1.5, 1.5, size=coords.shape)
    g = rng.normal(size=(E, 3, 3, h, w))
    lhs = (O.corr_index_forward(vol, coords, 1) * g).sum()
    rhs = (vol * O.corr_index_backward(vol.shape, coords, g, 1)).sum()
    np.testing.assert_allclose(lhs, rhs, rtol=1e-10)


def one_hot_window_case(r=3, h=9, w=11, y=4, x=3):
    """A hand-computed asymmetric fixture for the window's channel order: the only non-zero
    correlation of source pixel (y, x) is with target pixel (y, x + 2), and the pixel looks at
    its own position.  Offset (+2, 0) must light up window element (i, j) = (r + 2, r), i.e.
    channel (r + 2) * (2r + 1) + r = 38 for r = 3 (X offset slow), and nothing else."""
    vol = np.zeros((1, h, w, h, w))
    vol[0, y, x, y, x + 2] = 1.0
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    coords = np.stack([xx, yy])[None].astype(float)
    return vol, coords, (y, x), (r + 2) * (2 * r + 1) + r


def test_window_channel_order_one_hot():
    vol, coords, (y, x), channel = one_hot_window_case()
    assert channel == 38
    out = O.corr_index_forward(vol, coords, 3).reshape(1, 49, *vol.shape[1:3])
    lit = np.nonzero(out[0, :, y, x])[0]
    assert lit.tolist() == [channel] and out[0, channel, y, x] == 1.0
    # the on-the-fly form on feature maps whose only correlated pair is that same one
    f1 = np.zeros((1, 9, 11, 4)); f2 = np.zeros((1, 9, 11, 4))
    f1[0, y, x, 2] = 1.0
    f2[0, y, x + 2, 2] = 1.0
    alt = O.altcorr_forward(f1, f2, np.moveaxis(coords, 1, -1)[:, None], 3)    # [B,S,49,H,W]
    assert np.nonzero(alt[0, 0, :, y, x])[0].tolist() == [channel]


# ---------------------------------------------------------------------------
# the linear solve against the reference's own solvers (golden vectors)
# ---------------------------------------------------------------------------
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dba_solve_golden.npz")


def _golden_system(g, tag):
    H, E = g[f"{tag}_H"], g[f"{tag}_E"]
    return dict(H=H, E=E, C=g[f"{tag}_C"], v=g[f"{tag}_v"], w=g[f"{tag}_w"], P=H.shape[0],
                M=E.shape[1], HW=E.shape[3])


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("name", ["std", "stiff"])
def test_schur_solve_matches_reference_golden(tag, name):
    """float64 on both sides; 1e-9 relative to the largest entry."""
    g = np.load(GOLD)
    s = _golden_system(g, tag)
    ep, lm = g[f"{tag}_{name}_eplm"]
    S, rhs, E, Q = O.reduced_system(s, lm=lm, ep=ep)
    dx = O.solve_spd(S, rhs)
    dz = (Q * (s["w"].reshape(-1) - E.T @ dx)).reshape(s["M"], s["HW"])
    want_dx, want_dz = g[f"{tag}_{name}_dx"], g[f"{tag}_{name}_dz"]
    np.testing.assert_allclose(dx.reshape(-1, 6), want_dx, rtol=0, atol=1e-9 * np.abs(want_dx).max())
    np.testing.assert_allclose(dz, want_dz, rtol=0, atol=1e-9 * np.abs(want_dz).max())


@pytest.mark.parametrize("tag", ["a", "b"])
def test_motion_only_solve_and_the_twin_damping(tag):
    """The twin's block_solve (chol.py:32-43) multiplies its damping by a 6x6 identity that
    broadcasts over EVERY pose block, so off-diagonal blocks get ep + lm*H on their diagonals
    too.  The compiled extension the reference actually calls damps the diagonal of the
    assembled matrix, which is what the oracle does; with the twin's damping the oracle's
    solver reproduces the golden vector, which pins the factorisation itself."""
    g = np.load(GOLD)
    s = _golden_system(g, tag)
    P = s["P"]
    Hb = s["H"].copy()
    idx = np.arange(6)
    Hb[:, :, idx, idx] += 0.1 + 1e-4 * Hb[:, :, idx, idx]
    H = Hb.transpose(0, 2, 1, 3).reshape(6 * P, 6 * P)
    dx = O.solve_spd(H, s["v"].reshape(-1)).reshape(P, 6)
    want = g[f"{tag}_motion_dx"]
    np.testing.assert_allclose(dx, want, rtol=0, atol=1e-9 * np.abs(want).max())


def test_failed_factorisation_returns_zero_like_the_reference():
    g = np.load(GOLD)
    assert not g["bad_dx"].any()
    assert not O.solve_spd(-np.eye(12), np.ones(12)).any()


def test_altcorr_equals_lookup_in_the_all_pairs_volume():
    """The on-the-fly correlation window (modules/corr.py:74-125) restated position by position
    must equal the windowed lookup (corr.py:6-20) in the all-pairs volume (corr.py:63-71) of
    the same maps, and its backward must be the adjoint of its forward."""
    rng = np.random.default_rng(0)
    B, H, W, C, S, r = 2, 6, 7, 8, 2, 2
    f1 = rng.normal(size=(B, H, W, C)); f2 = rng.normal(size=(B, 5, 9, C))
    co = np.stack([rng.uniform(-3, 12, size=(B, S, H, W)), rng.uniform(-3, 8, size=(B, S, H, W))], -1)
    out = O.altcorr_forward(f1, f2, co, r)
    vol = np.einsum("bhwc,byxc->bhwyx", f1, f2)
    for s_ in range(S):
        ref = O.corr_index_forward(vol, np.moveaxis(co[:, s_], -1, 1), r)
        np.testing.assert_allclose(out[:, s_], ref.reshape(B, (2 * r + 1) ** 2, H, W), atol=1e-12)
    g = rng.normal(size=out.shape)
    g1, g2 = O.altcorr_backward(f1, f2, co, g, r)
    e1 = rng.normal(size=f1.shape); e2 = rng.normal(size=f2.shape); h = 1e-6
    num = (O.altcorr_forward(f1 + h * e1, f2 + h * e2, co, r) -
           O.altcorr_forward(f1 - h * e1, f2 - h * e2, co, r)) / (2 * h)
    np.testing.assert_allclose((num * g).sum(), (g1 * e1).sum() + (g2 * e2).sum(), rtol=1e-7)
