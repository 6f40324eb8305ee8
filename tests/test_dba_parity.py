"""GPU parity: HIP dense bundle adjustment (through the C ABI) against the numpy
float64 oracle.  Tolerance: 1e-4 relative (norm-wise) on pose and inverse-depth
updates (BASELINE.json north_star: "within 1e-4 rel on ... pose deltas").
"""
import numpy as np
import pytest
import torch

from oracle import dba_oracle as O
from vings_mono_amd import synth

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def to_dev(p, dev):
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)
    return dict(poses=f(p["poses"]), disps=f(p["disps"]), intr=f(p["intr"]),
                sens=f(p["disps_sens"]), target=f(p["target"]), weight=f(p["weight"]),
                eta=f(p["eta"]), ii=torch.from_numpy(p["ii"]).to(dev),
                jj=torch.from_numpy(p["jj"]).to(dev))


def f32(p):
    """The oracle sees exactly the float32-rounded inputs the GPU gets."""
    q = dict(p)
    for k in ("poses", "disps", "intr", "disps_sens", "target", "weight", "eta"):
        q[k] = p[k].astype(np.float32).astype(np.float64)
    return q


def rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30)


def pose_delta(new, old):
    return O.se3_mul(new, O.se3_inv(old))[:, :6]


CASES = [
    dict(n_frames=9, ht=43, wd=77, n_edges=48),          # reference-native grid
    dict(n_frames=6, ht=12, wd=16, n_edges=20, seed=5),
    dict(n_frames=9, ht=67, wd=120, n_edges=48, seed=11),
    dict(n_frames=7, ht=30, wd=41, n_edges=30, seed=3, buffer=20),
    # the grid bench.py times (1080p / 8): the Schur kernel keeps its accumulators over
    # 2048-pixel spans here, which the smaller grids above do not reach
    dict(n_frames=9, ht=135, wd=240, n_edges=48),
    # every ordered pair of 14 frames: 13 edges leave each frame, a depth variable couples 13
    # poses = 79 rows, past the 64 the Schur kernel keeps in registers (its wide path)
    dict(n_frames=14, ht=10, wd=13, n_edges=182, seed=21),
    # every ordered pair of 36 frames: more edges per frame than the linearisation caches in
    # LDS (32), and 210 unknowns, past what the solve keeps in LDS (128)
    dict(n_frames=36, ht=6, wd=9, n_edges=1260, seed=22),
    # fewer pixels than one 16-pixel step of the Schur kernel, two frames, one free pose
    dict(n_frames=2, ht=3, wd=4, n_edges=2, seed=23),
    dict(n_frames=3, ht=1, wd=17, n_edges=6, seed=24),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("itrs", [1, 2])
def test_ba_parity(case, itrs):
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    p = f32(synth.make_dba_problem(**case))
    d = to_dev(p, dev)
    dba.ba(d["poses"], d["disps"], d["intr"], d["sens"], d["target"], d["weight"], d["eta"],
           d["ii"], d["jj"], p["t0"], p["t1"], itrs, 1e-4, 0.1, False)
    torch.cuda.synchronize()
    rp, rd, rdx, _ = O.ba(p["poses"], p["disps"], p["intr"], p["disps_sens"], p["target"],
                          p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], itrs=itrs,
                          state_dtype=np.float32)
    got_p = d["poses"].cpu().numpy().astype(np.float64)
    got_d = d["disps"].cpu().numpy().astype(np.float64)
    n = case["n_frames"]
    dx, status = dba.last_update()
    assert status["depth_variables"] == p["eta"].shape[0]
    assert not status["cholesky_failed"] and not status["bad_index"]
    # the pose delta of the last iteration, read before it is rounded into float32 poses
    if itrs == 1:
        assert rel(dx.numpy(), rdx) < RTOL, rel(dx.numpy(), rdx)
    else:
        # The second update is taken at the state the first one left in the float32 buffers;
        # a one-ulp difference in that rounding moves the (100x smaller) second update by more
        # than 1e-4 of itself.  So iteration 2 is checked from the GPU's own intermediate
        # state: ba(itrs=1) twice on the GPU must equal ba(itrs=2), and the oracle started from
        # the GPU's state after iteration 1 must give the same second update.
        d1 = to_dev(p, dev)
        dba.ba(d1["poses"], d1["disps"], d1["intr"], d1["sens"], d1["target"], d1["weight"],
               d1["eta"], d1["ii"], d1["jj"], p["t0"], p["t1"], 1, 1e-4, 0.1, False)
        mid_p = d1["poses"].cpu().numpy().astype(np.float64)
        mid_d = d1["disps"].cpu().numpy().astype(np.float64)
        dba.ba(d1["poses"], d1["disps"], d1["intr"], d1["sens"], d1["target"], d1["weight"],
               d1["eta"], d1["ii"], d1["jj"], p["t0"], p["t1"], 1, 1e-4, 0.1, False)
        dx2, _ = dba.last_update()
        np.testing.assert_allclose(d1["poses"].cpu().numpy(), d["poses"].cpu().numpy(), atol=2e-7)
        np.testing.assert_allclose(d1["disps"].cpu().numpy(), d["disps"].cpu().numpy(), atol=2e-7)
        _, _, rdx2, _ = O.ba(mid_p, mid_d, p["intr"], p["disps_sens"], p["target"], p["weight"],
                             p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], itrs=1)
        assert rel(dx2.numpy(), rdx2) < RTOL, rel(dx2.numpy(), rdx2)
        rp, rd, _, _ = O.ba(mid_p, mid_d, p["intr"], p["disps_sens"], p["target"], p["weight"],
                            p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], itrs=1)
    # poses / depths after retraction: float32 storage bounds the agreement
    np.testing.assert_allclose(got_p[1:n], rp[1:n], atol=1e-6, rtol=2e-6)
    assert rel(got_d[:n] - p["disps"][:n], rd[:n] - p["disps"][:n]) < RTOL
    # gauge frame and unused buffer rows untouched
    np.testing.assert_array_equal(got_p[0], p["poses"][0].astype(np.float32))
    np.testing.assert_array_equal(got_p[n:], p["poses"][n:].astype(np.float32))
    np.testing.assert_array_equal(got_d[n:], p["disps"][n:].astype(np.float32))


def test_ba_motion_only_and_sensor_prior():
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    p = f32(synth.make_dba_problem(n_frames=6, ht=20, wd=24, n_edges=20, seed=2))
    d = to_dev(p, dev)
    dba.ba(d["poses"], d["disps"], d["intr"], d["sens"], d["target"], d["weight"], d["eta"],
           d["ii"], d["jj"], p["t0"], p["t1"], 2, 1e-4, 0.1, True)
    rp, rd, _, _ = O.ba(p["poses"], p["disps"], p["intr"], p["disps_sens"], p["target"],
                        p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], itrs=2,
                        motion_only=True)
    got = d["poses"].cpu().numpy().astype(np.float64)
    assert rel(pose_delta(got[1:6], p["poses"][1:6]), pose_delta(rp[1:6], p["poses"][1:6])) < RTOL
    np.testing.assert_array_equal(d["disps"].cpu().numpy(), p["disps"].astype(np.float32))
    # sensor prior on half of the frames
    p2 = dict(p)
    sens = np.zeros_like(p["disps"]); sens[:3] = (p["disps"][:3] * 1.05).astype(np.float32)
    p2["disps_sens"] = sens
    d = to_dev(p2, dev)
    dba.ba(d["poses"], d["disps"], d["intr"], d["sens"], d["target"], d["weight"], d["eta"],
           d["ii"], d["jj"], p["t0"], p["t1"], 1, 1e-4, 0.1, False)
    rp, rd, _, _ = O.ba(p2["poses"], p2["disps"], p2["intr"], p2["disps_sens"], p2["target"],
                        p2["weight"], p2["eta"], p2["ii"], p2["jj"], p2["t0"], p2["t1"], itrs=1)
    assert rel(d["disps"].cpu().numpy()[:6] - p["disps"][:6], rd[:6] - p["disps"][:6]) < RTOL


def test_ba_fixed_prefix_and_self_edge():
    """t0 = 3: frames 0..2 are fixed but their depths are still variables; one self edge
    exercises the stereo-baseline branch."""
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    p = synth.make_dba_problem(n_frames=8, ht=16, wd=20, n_edges=30, seed=8)
    p["ii"] = np.concatenate([p["ii"], [4]]); p["jj"] = np.concatenate([p["jj"], [4]])
    rng = np.random.default_rng(0)
    p["target"] = np.concatenate([p["target"], p["target"][:1] + rng.normal(size=p["target"][:1].shape)])
    p["weight"] = np.concatenate([p["weight"], p["weight"][:1]])
    p["t0"] = 3
    kx = np.unique(np.concatenate([np.arange(3, 8), p["ii"]]))
    p["eta"] = 1e-3 * rng.uniform(0.5, 1.5, size=(len(kx), 16, 20))
    p = f32(p)
    d = to_dev(p, dev)
    dba.ba(d["poses"], d["disps"], d["intr"], d["sens"], d["target"], d["weight"], d["eta"],
           d["ii"], d["jj"], p["t0"], p["t1"], 2, 1e-4, 0.1, False)
    rp, rd, _, _ = O.ba(p["poses"], p["disps"], p["intr"], p["disps_sens"], p["target"],
                        p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], itrs=2)
    got_p = d["poses"].cpu().numpy().astype(np.float64)
    assert rel(pose_delta(got_p[3:8], p["poses"][3:8]), pose_delta(rp[3:8], p["poses"][3:8])) < RTOL
    assert rel(d["disps"].cpu().numpy()[:8] - p["disps"][:8], rd[:8] - p["disps"][:8]) < RTOL
    np.testing.assert_array_equal(got_p[:3], p["poses"][:3].astype(np.float32))


def test_bacore_split_api():
    """hessian / hessian_expand fill caller-allocated CPU float64 tensors; retract(dx) with
    the oracle's dx reproduces ba()."""
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    p = f32(synth.make_dba_problem(n_frames=7, ht=24, wd=32, n_edges=30, seed=4))
    d = to_dev(p, dev)
    P = p["t1"] - p["t0"]
    core = dba.BACore()
    core.init(d["poses"], d["disps"], d["intr"], d["sens"], d["target"], d["weight"], d["eta"],
              d["ii"], d["jj"], p["t0"], p["t1"], 2, 1e-4, 0.1, False)
    H = torch.zeros((6 * P, 6 * P), dtype=torch.float64)
    v = torch.zeros(6 * P, dtype=torch.float64)
    # another problem of the same layout solved between init() and hessian(): the core owns its
    # workspace, so its linearised system is not overwritten
    other = to_dev(f32(synth.make_dba_problem(n_frames=7, ht=24, wd=32, n_edges=30, seed=40)), dev)
    dba.ba(other["poses"], other["disps"], other["intr"], other["sens"], other["target"],
           other["weight"], other["eta"], other["ii"], other["jj"], p["t0"], p["t1"], 1, 1e-4, 0.1,
           False)
    Q, E = core.hessian_expand(H, v)
    s = O.build_system(p["poses"], p["disps"], p["intr"], p["disps_sens"], p["target"],
                       p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"])
    S, rhs, Eref, Qref = O.reduced_system(s, 0.0, 0.0)
    assert rel(H.numpy(), S) < RTOL and rel(v.numpy(), rhs) < RTOL
    assert rel(Q.cpu().numpy().reshape(-1), Qref) < 1e-5
    HW = 24 * 32
    assert E.shape == (P + len(p["ii"]), 6, HW)
    # per-pose rows = coupling of pose i with its own depth map
    Ei_ref = np.stack([s["E"][k, np.searchsorted(s["kx"], k + p["t0"])] for k in range(P)])
    # remove edge contributions that target the same pose (self edges: none here)
    Eii = np.zeros_like(Ei_ref)
    for e in range(len(p["ii"])):
        a = p["ii"][e] - p["t0"]
        if 0 <= a < P:
            Eii[a] += s["Ei"][e]
    assert rel(E[:P].cpu().numpy(), Eii) < RTOL
    free_j = (p["jj"] >= p["t0"])
    assert rel(E[P:].cpu().numpy()[free_j], s["Ej"][free_j]) < RTOL
    # solve on the host (as the reference does with GTSAM) and retract
    Hd = S.copy(); Hd[np.diag_indices(6 * P)] += 0.1 + 1e-4 * np.diag(s["H"].transpose(0, 2, 1, 3).reshape(6 * P, 6 * P))
    dx = np.linalg.solve(Hd, rhs)
    core.retract(torch.from_numpy(dx))
    rp, rd, _, _ = O.ba(p["poses"], p["disps"], p["intr"], p["disps_sens"], p["target"],
                        p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], itrs=1)
    got_p = d["poses"].cpu().numpy().astype(np.float64)
    assert rel(pose_delta(got_p[1:7], p["poses"][1:7]), pose_delta(rp[1:7], p["poses"][1:7])) < RTOL
    assert rel(d["disps"].cpu().numpy()[:7] - p["disps"][:7], rd[:7] - p["disps"][:7]) < RTOL
    # second hessian() re-linearises at the moved state
    core.hessian(H, v)
    s2 = O.build_system(rp, rd, p["intr"], p["disps_sens"], p["target"], p["weight"], p["eta"],
                        p["ii"], p["jj"], p["t0"], p["t1"])
    S2, rhs2, _, _ = O.reduced_system(s2, 0.0, 0.0)
    assert rel(H.numpy(), S2) < 5 * RTOL


def test_zero_residual_fixed_point():
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    p = f32(synth.make_dba_problem(n_frames=5, ht=10, wd=12, n_edges=12, seed=9, noise_px=0.0))
    # recompute exact targets from the float32-rounded state
    c, _ = O.reproject(p["poses"], p["disps"], p["intr"], p["ii"], p["jj"])
    p["target"] = np.ascontiguousarray(c.transpose(0, 3, 1, 2))
    d = to_dev(p, dev)
    dba.ba(d["poses"], d["disps"], d["intr"], d["sens"], d["target"], d["weight"], d["eta"],
           d["ii"], d["jj"], p["t0"], p["t1"], 2, 1e-4, 0.1, False)
    np.testing.assert_allclose(d["poses"].cpu().numpy(), p["poses"], atol=2e-5)
    np.testing.assert_allclose(d["disps"].cpu().numpy(), p["disps"], atol=2e-4)


def test_frame_distance_parity():
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    p = f32(synth.make_dba_problem(n_frames=9, ht=43, wd=77, n_edges=48))
    n = 9
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    ii, jj = ii.reshape(-1), jj.reshape(-1)
    d = to_dev(p, dev)
    got = dba.frame_distance(d["poses"], d["disps"], d["intr"], torch.from_numpy(ii).to(dev),
                             torch.from_numpy(jj).to(dev), 0.3).cpu().numpy()
    ref = O.frame_distance(p["poses"], p["disps"], p["intr"], ii, jj, 0.3)
    np.testing.assert_allclose(got, ref, rtol=RTOL)


@pytest.mark.parametrize("half", [False, True])
def test_corr_index_parity(half):
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(6)
    E, h, w, r = 3, 11, 13, 3
    for lvl in range(3):
        h2, w2 = h >> lvl, w >> lvl
        vol = rng.normal(size=(E, h, w, h2, w2)).astype(np.float32)
        coords = np.stack([rng.uniform(-3, w2 + 2, size=(E, h, w)),
                           rng.uniform(-3, h2 + 2, size=(E, h, w))], 1).astype(np.float32)
        dt = torch.float16 if half else torch.float32
        tv = torch.from_numpy(vol).to(dev).to(dt)
        volr = tv.float().cpu().numpy()
        got, = dba.corr_index_forward(tv, torch.from_numpy(coords).to(dev), r)
        ref = O.corr_index_forward(volr, coords, r)
        tol = 2e-3 if half else 1e-5
        np.testing.assert_allclose(got.float().cpu().numpy(), ref, rtol=tol, atol=tol)
        g = rng.normal(size=ref.shape).astype(np.float32)
        tg = torch.from_numpy(g).to(dev).to(dt)
        gv, = dba.corr_index_backward(tv, torch.from_numpy(coords).to(dev), tg, r)
        refb = O.corr_index_backward(vol.shape, coords, tg.float().cpu().numpy(), r)
        tolb = 2e-2 if half else 1e-5
        np.testing.assert_allclose(gv.float().cpu().numpy(), refb, rtol=tolb, atol=tolb)


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("r", [3, 2])
def test_corr_index_forward_reference_grid(half, r):
    """The lookup at the reference's own grid (43 x 77: a pixel count that is no multiple of the
    64 pixels a wave takes) for radius 3 (cooperative kernel) and 2 (lane-per-pixel kernel);
    float tolerance 1e-5, half 2e-3 (output rounding)."""
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(16 + r)
    E, h, w = 2, 43, 77
    for lvl in (0, 2):
        h2, w2 = h >> lvl, w >> lvl
        vol = rng.normal(size=(E, h, w, h2, w2)).astype(np.float32)
        coords = np.stack([rng.uniform(-5, w2 + 4, size=(E, h, w)),
                           rng.uniform(-5, h2 + 4, size=(E, h, w))], 1).astype(np.float32)
        tv = torch.from_numpy(vol).to(dev).to(torch.float16 if half else torch.float32)
        got, = dba.corr_index_forward(tv, torch.from_numpy(coords).to(dev), r)
        ref = O.corr_index_forward(tv.float().cpu().numpy(), coords, r)
        tol = 2e-3 if half else 1e-5
        np.testing.assert_allclose(got.float().cpu().numpy(), ref, rtol=tol, atol=tol)


ALT_CASES = [
    dict(B=6, S=1, H=43, W=77, H2=43, W2=77, C=128, r=3),        # reference grid, level 0
    dict(B=3, S=2, H=43, W=77, H2=21, W2=38, C=128, r=3),        # a coarser pyramid level, S = 2
    dict(B=2, S=1, H=135, W=240, H2=135, W2=240, C=128, r=3),    # 1080p / 8: no volume possible
    dict(B=2, S=1, H=9, W=35, H2=7, W2=11, C=12, r=2),           # ragged chunk, small channels
    dict(B=1, S=1, H=5, W=6, H2=5, W2=6, C=4, r=4),              # window larger than a wave
    # smooth flow fields (what a tracker feeds): neighbouring windows overlap and the workgroup
    # stages their bounding rectangle through LDS; ragged pixel blocks, borders, S = 2
    dict(B=3, S=2, H=43, W=77, H2=43, W2=77, C=128, r=3, smooth=1.0),
    dict(B=2, S=1, H=135, W=240, H2=67, W2=120, C=128, r=3, smooth=0.5),
    dict(B=2, S=1, H=30, W=50, H2=30, W2=50, C=32, r=2, smooth=2.0),
    dict(B=2, S=1, H=18, W=21, H2=18, W2=21, C=16, r=0, smooth=0.3),
]


def _alt_inputs(c, seed=3):
    rng = np.random.default_rng(seed)
    f1 = rng.normal(size=(c["B"], c["H"], c["W"], c["C"])).astype(np.float32) / 4
    f2 = rng.normal(size=(c["B"], c["H2"], c["W2"], c["C"])).astype(np.float32) / 4
    # targets scattered over fmap2 and a margin outside it (zero padding), a few far away
    x = rng.uniform(-5, c["W2"] + 4, size=(c["B"], c["S"], c["H"], c["W"]))
    y = rng.uniform(-5, c["H2"] + 4, size=(c["B"], c["S"], c["H"], c["W"]))
    if c.get("smooth"):
        # pixel grid scaled onto fmap2 + a per-(b, s) translation and shear that pushes part of
        # the field across the border + per-pixel jitter
        gy, gx = np.meshgrid(np.arange(c["H"]), np.arange(c["W"]), indexing="ij")
        sh = rng.uniform(-6, 6, size=(c["B"], c["S"], 2, 1, 1))
        x = gx * (c["W2"] / c["W"]) + sh[:, :, 0] + 0.03 * gy + rng.normal(size=x.shape) * c["smooth"]
        y = gy * (c["H2"] / c["H"]) + sh[:, :, 1] - 0.02 * gx + rng.normal(size=y.shape) * c["smooth"]
        x[-1, -1, 2, 3] = -3e6; y[-1, -1, 1, 1] = c["H2"] + 40.0   # one stray target in a block
    x[0, 0, 0, 0] = 1e7; y[0, 0, 0, 1] = -1e7
    return f1, f2, np.stack([x, y], -1).astype(np.float32)


@pytest.mark.parametrize("case", ALT_CASES)
def test_altcorr_forward_parity(case):
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    f1, f2, co = _alt_inputs(case)
    got, = dba.altcorr_forward(torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev),
                               torch.from_numpy(co).to(dev), case["r"])
    want = O.altcorr_forward(f1, f2, co, case["r"])
    assert got.shape == want.shape
    assert rel(got.cpu().numpy(), want) < 1e-5
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=1e-5 * np.abs(want).max() + 1e-6)


@pytest.mark.parametrize("case", ALT_CASES[:2] + ALT_CASES[3:5])
def test_altcorr_backward_parity(case):
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    f1, f2, co = _alt_inputs(case, seed=4)
    rd2 = (2 * case["r"] + 1) ** 2
    g = np.random.default_rng(5).normal(size=(case["B"], case["S"], rd2, case["H"], case["W"])).astype(np.float32)
    g1, g2, gc = dba.altcorr_backward(torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev),
                                      torch.from_numpy(co).to(dev), torch.from_numpy(g).to(dev), case["r"])
    w1, w2 = O.altcorr_backward(f1, f2, co, g, case["r"])
    assert rel(g1.cpu().numpy(), w1) < 1e-5 and rel(g2.cpu().numpy(), w2) < 1e-5
    assert gc.shape == co.shape and float(gc.abs().max()) == 0.0


def test_window_channel_order_one_hot_on_the_gpu():
    """The hand-computed asymmetric fixture of tests/test_oracle_dba.py through both lookups: a
    target two pixels to the right lights up channel (r + 2) * 7 + r = 38 and nothing else."""
    from test_oracle_dba import one_hot_window_case
    from vings_mono_amd import dba
    vol, coords, (y, x), channel = one_hot_window_case()
    dev = torch.device("cuda:0")
    for dt in (torch.float32, torch.float16):
        (out,) = dba.corr_index_forward(torch.from_numpy(vol).to(dev, dt),
                                        torch.from_numpy(coords).float().to(dev), 3)
        win = out.float().reshape(49, *vol.shape[1:3])[:, y, x].cpu().numpy()
        assert np.nonzero(win)[0].tolist() == [channel] and win[channel] == 1.0
    f1 = torch.zeros((1, 9, 11, 16), device=dev); f2 = torch.zeros((1, 9, 11, 16), device=dev)
    f1[0, y, x, 2] = 1.0
    f2[0, y, x + 2, 2] = 1.0
    co = torch.from_numpy(np.moveaxis(coords, 1, -1)[:, None].copy()).float().to(dev)
    (alt,) = dba.altcorr_forward(f1, f2, co, 3)
    win = alt[0, 0, :, y, x].cpu().numpy()
    assert np.nonzero(win)[0].tolist() == [channel] and win[channel] == 1.0
    # the general-radius kernels follow the same order
    (out2,) = dba.corr_index_forward(torch.from_numpy(vol).float().to(dev), torch.from_numpy(coords).float().to(dev), 2)
    assert np.nonzero(out2.reshape(25, *vol.shape[1:3])[:, y, x].cpu().numpy())[0].tolist() == [4 * 5 + 2]


def test_altcorr_equals_volume_lookup_on_the_gpu():
    """altcorr_forward and corr_index_forward(all-pairs volume) are two routes to one window."""
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    c = dict(B=4, S=1, H=24, W=32, H2=24, W2=32, C=128, r=3)
    f1, f2, co = _alt_inputs(c, seed=8)
    t1, t2, tc = (torch.from_numpy(a).to(dev) for a in (f1, f2, co))
    alt, = dba.altcorr_forward(t1, t2, tc, 3)
    vol = torch.einsum("bhwc,byxc->bhwyx", t1, t2).contiguous()
    ref, = dba.corr_index_forward(vol, tc[:, 0].permute(0, 3, 1, 2).contiguous(), 3)
    assert rel(alt[:, 0].cpu().numpy(), ref.reshape(4, 49, 24, 32).cpu().numpy().astype(np.float64)) < 1e-5


def test_altcorr_full_grid_properties():
    """The 1080p / 8 tracker grid with 12 edges and a smooth flow field (the size at which no
    all-pairs volume exists and the oracle takes minutes): the window is linear in either
    feature map, and translating the second map together with the targets by whole pixels
    leaves every window value that stays inside both maps unchanged."""
    from vings_mono_amd import dba
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(17)
    E, H, W, C, r = 12, 135, 240, 128, 3
    f1 = (torch.randn(E, H, W, C, generator=g) / 4).to(dev)
    f2 = (torch.randn(E, H, W, C, generator=g) / 4).to(dev)
    h2 = (torch.randn(E, H, W, C, generator=g) / 4).to(dev)
    xy = torch.stack(torch.meshgrid(torch.arange(W, dtype=torch.float32),
                                    torch.arange(H, dtype=torch.float32), indexing="xy"), -1)
    xy = xy[None, None].repeat(E, 1, 1, 1, 1) + (torch.rand(E, 1, 1, 1, 2, generator=g) - 0.5) * 20
    xy = (xy + torch.randn(xy.shape, generator=g) * 0.5).to(dev).contiguous()
    a, = dba.altcorr_forward(f1, f2, xy, r)
    b, = dba.altcorr_forward(f1, h2, xy, r)
    c, = dba.altcorr_forward(f1, 0.75 * f2 - 1.5 * h2, xy, r)
    want = 0.75 * a - 1.5 * b
    assert float((c - want).abs().max()) <= 2e-5 * float(want.abs().max())
    d, = dba.altcorr_forward(2.0 * f1, f2, xy, r)
    assert torch.equal(d, 2.0 * a)                        # scaling by two is exact in float32
    # shift the second map 5 right, 3 down (zero fill) and the targets with it
    sh = torch.zeros_like(f2)
    sh[:, 3:, 5:] = f2[:, :-3, :-5]
    e, = dba.altcorr_forward(f1, sh, xy + torch.tensor([5.0, 3.0], device=dev), r)
    x0 = torch.floor(xy[..., 0]) - r
    y0 = torch.floor(xy[..., 1]) - r
    inside = ((x0 >= 0) & (x0 + 2 * r + 1 < W - 5) & (y0 >= 0) & (y0 + 2 * r + 1 < H - 3))[:, :, None]
    assert float(inside.float().mean()) > 0.7
    assert float(((e - a).abs() * inside).max()) <= 2e-5 * float(a.abs().max())


def test_schur_grid_does_not_change_the_update(tmp_path):
    """The Schur kernel's launch shape (workgroups per depth variable, pixels per wave) is a
    tuning choice: far fewer and far more workgroups than the default must give the same pose
    update up to the order of the float64 additions.  The target is read once per process, so
    each setting runs in a process of its own."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from vings_mono_amd import dba, synth\n"
        "p = synth.make_dba_problem(n_frames=9, ht=67, wd=120, n_edges=48, seed=11)\n"
        "f = lambda a: torch.from_numpy(np.asarray(a)).float().cuda().contiguous()\n"
        "ii, jj = torch.from_numpy(p['ii']).cuda(), torch.from_numpy(p['jj']).cuda()\n"
        "poses, disps = f(p['poses']), f(p['disps'])\n"
        "dba.ba(poses, disps, f(p['intr']), f(p['disps_sens']), f(p['target']), f(p['weight']), f(p['eta']),\n"
        "       ii, jj, p['t0'], p['t1'], 1, 1e-4, 0.1, False)\n"
        "dx, st = dba.last_update()\n"
        "assert not st['cholesky_failed']\n"
        "np.save(sys.argv[1], dx.numpy())\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for wgs in ("9", "", "4096"):
        out = str(tmp_path / f"dx_{wgs or 'default'}.npy")
        env = dict(os.environ)
        env.pop("VINGS_SCHUR_WGS", None)
        if wgs:
            env["VINGS_SCHUR_WGS"] = wgs
        res = subprocess.run([sys.executable, "-c", code, out], env=env, capture_output=True, text=True,
                             timeout=300)
        assert res.returncode == 0, res.stderr[-3000:]
        outs.append(np.load(out))
    for other in (outs[0], outs[2]):
        assert rel(other, outs[1]) < 1e-10, rel(other, outs[1])
