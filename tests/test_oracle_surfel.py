"""CPU checks of the surfel oracle itself (no GPU).

The reference ships no golden vectors for the rasteriser (parity unpinned), so
the oracle is validated by closed-form invariants and by central finite
differences of its own forward in float64.
"""
import numpy as np
import pytest

from scenes import oracle_forward, scene


def test_forward_invariants(oracle32):
    sc = scene(3000, 160, 96, with_flow=True)
    st = oracle_forward(oracle32, sc)
    assert st["D"] == int(st["tiles_touched"].sum())
    # alpha channel is 1 - final transmittance, accumulated weights never exceed 1
    np.testing.assert_allclose(st["allmap"][1], 1.0 - st["final_T"][0], rtol=0, atol=1e-6)
    assert st["allmap"][1].max() <= 1.0 and st["allmap"][1].min() >= 0.0
    # sorted keys ascend; every tile range covers exactly the keys of that tile
    keys = st["keys"]
    assert np.all(keys[1:] >= keys[:-1])
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    for t, (a, b) in enumerate(st["ranges"]):
        assert np.all(tiles[a:b] == t)
    assert int((st["ranges"][:, 1] - st["ranges"][:, 0]).sum()) == st["D"]
    # stable: equal keys keep splat order
    same = keys[1:] == keys[:-1]
    assert np.all(st["vals"][1:][same] > st["vals"][:-1][same])
    # culled splats have no footprint
    assert np.all(st["tiles_touched"][st["radii"] == 0] == 0)


def test_single_splat_closed_form(oracle64):
    """A fronto-parallel splat centred on a pixel: alpha at the centre is the opacity,
    depth is its z, normal faces the camera."""
    W, H = 64, 48
    sc = scene(1, W, H)
    fx = 0.6 * W
    z = 4.0
    sc["means3D"] = np.array([[0.0, 0.0, z]], np.float32)
    sc["rotations"] = np.array([[1.0, 0, 0, 0]], np.float32)
    sc["scales"] = np.array([[0.5, 0.5]], np.float32)
    sc["opacities"] = np.array([[0.7]], np.float32)
    sc["colors"] = np.array([[0.2, 0.4, 0.6]], np.float32)
    st = oracle_forward(oracle64, sc)
    # pixel centre of the principal point is ((W-1)/2, (H-1)/2): not an integer pixel, so
    # evaluate the analytic falloff at the nearest pixel instead
    px, py = W // 2, H // 2
    du = (px - (W - 1) / 2.0) / fx * z / 0.5
    dv = (py - (H - 1) / 2.0) / fx * z / 0.5
    alpha = 0.7 * np.exp(-0.5 * (du * du + dv * dv))
    # the projection matrix is stored in float32, hence 1e-6
    assert abs(st["allmap"][1, py, px] - alpha) < 1e-6
    assert abs(st["allmap"][0, py, px] / st["allmap"][1, py, px] - z) < 1e-6
    n = st["allmap"][2:5, py, px] / alpha
    np.testing.assert_allclose(n, [0, 0, -1], atol=1e-6)
    np.testing.assert_allclose(st["color"][:, py, px], alpha * sc["colors"][0].astype(np.float64),
                               atol=1e-6)


def test_sort_is_depth_permutation_invariant(oracle32):
    sc = scene(500, 96, 64)
    st0 = oracle_forward(oracle32, sc)
    perm = np.random.default_rng(1).permutation(500)
    sc2 = dict(sc)
    for k in ("means3D", "scales", "rotations", "opacities", "colors"):
        sc2[k] = sc[k][perm]
    st1 = oracle_forward(oracle32, sc2)
    np.testing.assert_allclose(st0["color"], st1["color"], atol=2e-6)
    np.testing.assert_allclose(st0["allmap"][[0, 1, 6]], st1["allmap"][[0, 1, 6]], atol=2e-5)


def test_pixel_mask_gates_pixels(oracle32):
    sc = scene(800, 64, 48, bg=(0.1, 0.2, 0.3))
    mask = np.ones((48, 64), bool)
    mask[:24] = False
    st = oracle_forward(oracle32, sc, pixel_mask=mask)
    full = oracle_forward(oracle32, sc)
    assert np.all(st["allmap"][:, :24] == 0)
    np.testing.assert_allclose(st["color"][:, :24], np.broadcast_to(sc["bg"][:, None, None], (3, 24, 64)))
    np.testing.assert_array_equal(st["color"][:, 24:], full["color"][:, 24:])
    g = oracle32.backward(st, np.ones((3, 48, 64)), np.ones((9, 48, 64)))
    gf = oracle32.backward(full, np.ones((3, 48, 64)), np.ones((9, 48, 64)))
    assert np.abs(g["scores"][:, 0]).sum() < np.abs(gf["scores"][:, 0]).sum()


@pytest.mark.parametrize("scale_mul,label", [(3.0, "ray-splat branch"), (0.08, "low-pass branch")])
def test_backward_matches_finite_differences(oracle64, scale_mul, label):
    W, H, N = 48, 32, 24
    rng = np.random.default_rng(0)
    sc = scene(N, W, H, z_range=(2.0, 6.0), scale_mul=scale_mul, with_flow=True, bg=(0.3, 0.1, 0.2))
    a = {k: sc[k].astype(np.float64) for k in ("means3D", "scales", "rotations", "opacities", "colors")}
    a["opacities"] = a["opacities"] * 0.5 + 0.2
    view, proj, bg = (sc[k].astype(np.float64) for k in ("view", "proj", "bg"))
    flow = sc["flow"].astype(np.float64)
    wC = rng.normal(size=(3, H, W))
    wA = rng.normal(size=(9, H, W))
    wA[5] = 0  # median depth is piecewise constant in the inputs

    def run(a):
        return oracle64.forward(a["means3D"], a["scales"], a["rotations"], a["opacities"],
                                a["colors"], view, proj, bg, H, W, flow=flow)

    def loss(a):
        st = run(a)
        return (st["color"] * wC).sum() + (st["allmap"] * wA).sum()

    st = run(a)
    assert st["D"] > 0 and st["n_contrib"][0].max() > 0
    g = oracle64.backward(st, wC, wA)
    names = {"means3D": "means3D", "scales": "scales", "rotations": "rots",
             "opacities": "opacity", "colors": "colors"}
    for k, gk in names.items():
        x = a[k]
        an = g[gk].reshape(x.shape)
        fd = np.zeros_like(x)
        it = np.nditer(x, flags=["multi_index"])
        for _ in it:
            idx = it.multi_index
            old = x[idx]
            h = 1e-6
            x[idx] = old + h
            lp = loss(a)
            x[idx] = old - h
            lm = loss(a)
            x[idx] = old
            fd[idx] = (lp - lm) / (2 * h)
        err = np.abs(fd - an) / (np.abs(fd) + np.abs(an) + 1e-8)
        assert err.max() < 1e-4, (label, k, err.max())


def test_scores_definition(oracle32):
    """scores[:,0] is the blend-weight mass of a splat and accumulates under a zero
    upstream gradient (the reference back-propagates (rgb*0).sum() and reads it,
    /root/reference/scripts/loop/loop_model.py:185-186)."""
    sc = scene(1500, 96, 64)
    st = oracle_forward(oracle32, sc)
    g = oracle32.backward(st, np.zeros((3, 64, 96)), np.zeros((9, 64, 96)))
    assert np.all(g["scores"][:, 1] == 0)
    total_w = g["scores"][:, 0].sum()
    np.testing.assert_allclose(total_w, st["allmap"][1].sum(), rtol=1e-4)
    for k in ("means3D", "scales", "rots", "opacity", "colors"):
        assert np.all(g[k] == 0)


def test_sparse_adam_matches_formula(oracle32):
    rng = np.random.default_rng(3)
    N, M = 257, 3
    p = rng.normal(size=(N, M)).astype(np.float32)
    g = rng.normal(size=(N, M)).astype(np.float32)
    m = rng.normal(size=(N, M)).astype(np.float32) * 0.1
    v = np.abs(rng.normal(size=(N, M)).astype(np.float32)) * 0.01
    vis = rng.uniform(size=N) > 0.4
    p0, m0, v0 = p.copy(), m.copy(), v.copy()
    oracle32.sparse_adam(p, g, m, v, vis, 1e-2)
    m_ref = 0.9 * m0.astype(np.float64) + 0.1 * g
    v_ref = 0.999 * v0.astype(np.float64) + 0.001 * g.astype(np.float64) ** 2
    p_ref = p0 - 1e-2 * m_ref / (np.sqrt(v_ref) + 1e-15)
    np.testing.assert_allclose(p[vis], p_ref[vis], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(p[~vis], p0[~vis])
    np.testing.assert_array_equal(m[~vis], m0[~vis])
    np.testing.assert_array_equal(v[~vis], v0[~vis])
