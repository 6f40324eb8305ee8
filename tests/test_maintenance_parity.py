"""Map maintenance on the device (forward-only importance, stream compaction, pruning)
against the CPU oracle / numpy.  Importance: 1e-4 relative L2 against the oracle's
scores[:,0] (the backward pass' definition of the same quantity).  Compaction: bit-exact
against numpy boolean indexing."""
import numpy as np
import pytest
import torch

from scenes import oracle_forward, scene
from vings_mono_amd import harness, maintenance, surfel, synth
import test_surfel_parity as TS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("case", [dict(n=10000, W=640, H=480),
                                  dict(n=4000, W=250, H=131, scale_mul=2.0),
                                  dict(n=3000, W=200, H=120, scale_mul=0.05)])
def test_importance_matches_backward_scores(case, oracle32):
    dev = torch.device(DEV)
    sc = scene(**case)
    H, W = sc["H"], sc["W"]
    ref_st = oracle_forward(oracle32, sc)
    zc, za = np.zeros((3, H, W), np.float32), np.zeros((9, H, W), np.float32)
    want = oracle32.backward(ref_st, zc, za)["scores"][:, 0]
    inp = TS._dev_inputs(sc, dev)
    got, radii = surfel.view_importance(inp["means3D"], inp["colors"], inp["opacities"],
                                        inp["scales"], inp["rotations"], TS._settings(sc, dev))
    got = got.cpu().numpy().astype(np.float64)
    rel = np.sqrt(((got - want) ** 2).sum()) / np.sqrt((want.astype(np.float64) ** 2).sum())
    assert rel <= 1e-4, rel
    assert np.all(got[want == 0] == 0)
    np.testing.assert_array_equal(radii.cpu().numpy(), ref_st["radii"])
    # accumulate = True adds a second view on top
    out = torch.from_numpy(got.astype(np.float32)).to(dev)
    surfel.view_importance(inp["means3D"], inp["colors"], inp["opacities"], inp["scales"],
                           inp["rotations"], TS._settings(sc, dev), out=out, accumulate=True)
    np.testing.assert_allclose(out.cpu().numpy(), 2 * got, rtol=1e-5, atol=1e-6)


def test_importance_honours_pixel_mask(oracle32):
    dev = torch.device(DEV)
    sc = scene(3000, 160, 96)
    mask = np.random.default_rng(5).uniform(size=(96, 160)) > 0.4
    ref_st = oracle_forward(oracle32, sc, pixel_mask=mask)
    z = lambda c: np.zeros((c, 96, 160), np.float32)
    want = oracle32.backward(ref_st, z(3), z(9))["scores"][:, 0]
    inp = TS._dev_inputs(sc, dev)
    got, _ = surfel.view_importance(inp["means3D"], inp["colors"], inp["opacities"], inp["scales"],
                                    inp["rotations"], TS._settings(sc, dev, mask))
    got = got.cpu().numpy().astype(np.float64)
    assert np.sqrt(((got - want) ** 2).sum()) <= 1e-4 * np.sqrt((want.astype(np.float64) ** 2).sum())


@pytest.mark.parametrize("n,frac", [(1, 1.0), (1, 0.0), (257, 0.5), (100000, 0.9), (4096, 0.0),
                                    (5000, 1.0)])
def test_compaction_is_bit_exact(n, frac):
    rng = np.random.default_rng(n + int(frac * 10))
    keep = rng.random(n) < frac
    host = [rng.normal(size=(n, 3)).astype(np.float32), rng.normal(size=(n, 1)).astype(np.float32),
            rng.normal(size=(n,)).astype(np.float32), rng.integers(-5, 1 << 40, n).astype(np.int64),
            rng.random(n) < 0.5, rng.normal(size=(n, 4)).astype(np.float32),
            rng.normal(size=(n, 2)).astype(np.float32), rng.integers(0, 255, (n, 3)).astype(np.uint8)]
    outs, m = maintenance.compact([torch.from_numpy(a).to(DEV) for a in host],
                                  torch.from_numpy(keep).to(DEV))
    assert m == int(keep.sum())
    for got, a in zip(outs, host):
        np.testing.assert_array_equal(got.cpu().numpy(), a[keep])


def test_prune_keeps_parameters_state_and_records():
    dev = torch.device(DEV)
    W, H = 160, 96
    raw = synth.make_splats(4000, W, H, n_views=2, seed=8)
    intr = synth.intrinsic_for(W, H)
    w2c = torch.linalg.inv(synth.window_poses(2)[1]).to(dev)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    with torch.no_grad():
        pr = harness.render(gtp, w2c, intr)
        gt = {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(),
              "depth_cov": torch.ones_like(pr["depth"])}
    m = harness.SurfelMapper(raw, dev)
    cam = harness.view_settings(w2c, intr)
    m.fused_view(cam, intr, gt, w2c[:3, :3].cpu(), 0)          # creates Adam state and scores
    before = {k: m.params[k].detach().clone() for k in m.NAMES}
    mom = {k: m.optimizer.state[m.params[k]]["exp_avg"].clone() for k in m.NAMES}
    glob = m.global_scores.clone()
    prune_mask = torch.rand(m.n, device=dev) < 0.3
    left = maintenance.prune(m, prune_mask)
    keep = ~prune_mask
    assert left == int(keep.sum()) == m.n
    for k in m.NAMES:
        assert torch.equal(m.params[k].detach(), before[k][keep])
        assert torch.equal(m.optimizer.state[m.params[k]]["exp_avg"], mom[k][keep])
        assert m.optimizer.state[m.params[k]]["exp_avg_sq"].shape == before[k][keep].shape
    assert torch.equal(m.global_scores, glob[keep])
    assert m.stable_mask.shape == (left,) and m.globalkf_id.shape == (left,)
    loss = m.fused_view(cam, intr, gt, w2c[:3, :3].cpu(), 1)   # the pruned map still trains
    assert torch.isfinite(loss).all()


def test_top3_update_matches_reference_logic():
    from oracle import mapper_oracle as mo
    from vings_mono_amd import _lib
    L = _lib.lib()
    n = 5000
    rng = np.random.default_rng(2)
    s = np.zeros((n, 3), np.float32)
    k = np.zeros((n, 3), np.int32)
    ts, tk = torch.from_numpy(s).to(DEV), torch.from_numpy(k).to(DEV)
    for kf in range(7):
        cur = (rng.random(n) * (rng.random(n) > 0.3)).astype(np.float32)
        cur[: n // 50] = 0.25                       # repeated value: ties must not insert
        s, k = mo.top3_update(s, k, cur, kf + 10)
        tc = torch.from_numpy(cur).to(DEV)
        _lib.check(L.vings_splat_top3_update(n, _lib.ptr(tc), kf + 10, _lib.ptr(ts), _lib.ptr(tk),
                                             _lib.stream_ptr(torch.device(DEV))), "top3")
    np.testing.assert_array_equal(ts.cpu().numpy(), s)
    np.testing.assert_array_equal(tk.cpu().numpy(), k)
    # the gated form: a view whose gate word lies below the threshold changes nothing, one at or
    # above it is folded in like the plain call does (loop_model.py:181 decided on the device)
    cur = rng.random(n).astype(np.float32)
    tc = torch.from_numpy(cur).to(DEV)
    gate = torch.tensor([0.05], device=DEV)
    _lib.check(L.vings_splat_top3_update_gated(n, _lib.ptr(tc), 99, _lib.ptr(ts), _lib.ptr(tk),
                                               _lib.ptr(gate), 0.1, _lib.stream_ptr(torch.device(DEV))),
               "top3 gated")
    np.testing.assert_array_equal(ts.cpu().numpy(), s)
    np.testing.assert_array_equal(tk.cpu().numpy(), k)
    gate.fill_(0.1)
    _lib.check(L.vings_splat_top3_update_gated(n, _lib.ptr(tc), 99, _lib.ptr(ts), _lib.ptr(tk),
                                               _lib.ptr(gate), 0.1, _lib.stream_ptr(torch.device(DEV))),
               "top3 gated")
    s, k = mo.top3_update(s, k, cur, 99)
    np.testing.assert_array_equal(ts.cpu().numpy(), s)
    np.testing.assert_array_equal(tk.cpu().numpy(), k)


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_top3_merge_of_rank_records_equals_the_sequential_pass(ranks):
    """The keyframes dealt out over `ranks` in contiguous blocks, one record per block, merged:
    bit for bit the record of the sequential pass (loop_model.py:190-211), ties (scores on a
    coarse grid: most splats see equal scores from several keyframes) and empty slots included."""
    from oracle import mapper_oracle as mo
    from vings_mono_amd import _lib
    L = _lib.lib()
    n, K = 4000, 13
    rng = np.random.default_rng(5 + ranks)
    curs = [(np.round(rng.random(n) * 6) / 6 * (rng.random(n) > 0.4)).astype(np.float32) for _ in range(K)]
    s, k = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.int32)
    for kf in range(K):
        s, k = mo.top3_update(s, k, curs[kf], kf + 3)
    per = -(-K // ranks)
    stream = _lib.stream_ptr(torch.device(DEV))
    all_s = torch.zeros((ranks, n, 3), dtype=torch.float32, device=DEV)
    all_k = torch.zeros((ranks, n, 3), dtype=torch.int32, device=DEV)
    for r in range(ranks):
        for kf in range(r * per, min(K, (r + 1) * per)):
            tc = torch.from_numpy(curs[kf]).to(DEV)
            _lib.check(L.vings_splat_top3_update(n, _lib.ptr(tc), kf + 3, _lib.ptr(all_s[r]),
                                                 _lib.ptr(all_k[r]), stream), "top3")
    out_s = torch.empty((n, 3), dtype=torch.float32, device=DEV)
    out_k = torch.empty((n, 3), dtype=torch.int32, device=DEV)
    _lib.check(L.vings_splat_top3_merge(n, ranks, _lib.ptr(all_s), _lib.ptr(all_k), _lib.ptr(out_s),
                                        _lib.ptr(out_k), stream), "top3 merge")
    np.testing.assert_array_equal(out_s.cpu().numpy(), s)
    np.testing.assert_array_equal(out_k.cpu().numpy(), k)
    assert (s[:, 0] == s[:, 1]).sum() == (s[:, 0] == 0).sum()       # ties never entered twice


def test_calc_score_and_rectify_gaussians():
    from oracle import mapper_oracle as mo
    dev = torch.device(DEV)
    W, H, K = 160, 96, 4
    raw = synth.make_splats(6000, W, H, n_views=K, seed=12)
    intr = synth.intrinsic_for(W, H)
    c2ws = synth.window_poses(K)
    m = harness.SurfelMapper(raw, dev)
    scores3, ids3 = maintenance.calc_score(m, [c2ws[k] for k in range(K)], list(range(K)), intr)
    # per-view importance through the backward route, then the oracle's top-3 bookkeeping
    s = np.zeros((m.n, 3), np.float32)
    ids = np.zeros((m.n, 3), np.int32)
    for k in range(K):
        sc = torch.zeros((m.n, 2), device=dev, requires_grad=True)
        rgb, _, _ = harness.render_raw(m.params, torch.linalg.inv(c2ws[k]).to(dev), intr, scores=sc)
        (rgb * 0).sum().backward()
        s, ids = mo.top3_update(s, ids, sc.grad[:, 0].cpu().numpy(), k)
    got_s, got_i = scores3.cpu().numpy(), ids3.cpu().numpy()
    np.testing.assert_allclose(got_s, s, rtol=1e-4, atol=1e-5)     # float32 sums, two orders
    assert (got_i == ids).mean() >= 0.999          # near-ties between views may swap

    # re-anchor: a drifted loop over keyframes 1..2 with per-keyframe scales
    new_c2ws = c2ws.clone()
    a = np.radians(3.0)
    Rz = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]],
                      dtype=torch.float32)
    for k in (1, 2):
        new_c2ws[k, :3, :3] = Rz @ c2ws[k, :3, :3]
        new_c2ws[k, :3, 3] += torch.tensor([0.3, -0.1, 0.2]) * k
    scales = torch.tensor([1.05, 0.97])
    xyz0 = m.params["_xyz"].detach().cpu().numpy()
    rot0 = m.params["_rotation"].detach().cpu().numpy()
    want_xyz, want_R = mo.reanchor(xyz0, rot0, got_i[:, 0], 1, 2, c2ws.numpy(), new_c2ws.numpy(),
                                   scales.numpy())
    keep = got_s[:, 0] >= 0.1
    n0 = m.n
    deleted = maintenance.rectify_gaussians(m, c2ws, new_c2ws, scales, 1, 2, scores3, ids3)
    assert np.array_equal(deleted.cpu().numpy(), ~keep) and m.n == int(keep.sum()) < n0
    got_xyz = m.params["_xyz"].detach().cpu().numpy()
    got_R = mo.quat_to_matrix_wxyz(m.params["_rotation"].detach().cpu().numpy().astype(np.float64))
    np.testing.assert_allclose(got_xyz, want_xyz[keep], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(got_R, want_R[keep], atol=5e-6)
    moved = ((got_i[:, 0] >= 1) & (got_i[:, 0] <= 2))[keep]
    assert moved.any() and not moved.all()
    assert np.array_equal(got_xyz[~moved], xyz0[keep][~moved])      # others untouched


def test_storage_control_prunes_by_forward_importance():
    dev = torch.device(DEV)
    W, H = 160, 96
    raw = synth.make_splats(6000, W, H, n_views=3, seed=4)
    intr = synth.intrinsic_for(W, H)
    w2cs = [torch.linalg.inv(p).to(dev) for p in synth.window_poses(3)]
    m = harness.SurfelMapper(raw, dev)
    m.stable_mask[::5] = True
    # the reference's route: render + backward per keyframe, read scores.grad[:,0]
    want = torch.zeros(m.n, device=dev)
    for w2c in w2cs:
        scores = torch.zeros((m.n, 2), device=dev, requires_grad=True)
        rgb, _, _ = harness.render_raw(m.params, w2c, intr, scores=scores)
        (rgb * 0).sum().backward()
        want += scores.grad[:, 0]
    got = maintenance.importance_over_views(m, w2cs, intr)
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    expect = (want > 0.05) & (~m.stable_mask) & (want < 0.8)
    n0 = m.n
    mask = maintenance.storage_control(m, w2cs, intr)
    # only splats whose score sits within rounding of a threshold may be classified differently
    assert int((mask != expect).sum()) <= 2
    assert m.n == n0 - int(mask.sum())
    maintenance.stablemask_control(m)
    assert float(m.local_scores.abs().max()) == 0.0


def test_knn3_mean_dist2_matches_reference_golden():
    """Against outputs of the reference's distCUDA2 (k-d tree, float32 sums) on 3000 points;
    float tolerance 1e-5 relative (the summation order of three squares differs)."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                "harness_golden.npz"))
    pts = torch.from_numpy(gold["knn_points"]).to(DEV)
    got = maintenance.mean_dist2_to_3nn(pts).cpu().numpy()
    np.testing.assert_allclose(got, gold["knn_dist2"], rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("n", [4, 257, 5000])
def test_knn3_mean_dist2_brute_force(n):
    g = torch.Generator().manual_seed(n)
    pts = torch.rand(n, 3, generator=g)
    pts[n // 2] = pts[0]                                  # a duplicate: distance 0 counts
    d = torch.cdist(pts.double(), pts.double()) ** 2
    d.fill_diagonal_(float("inf"))
    want = d.topk(3, dim=1, largest=False).values.mean(1).float()
    got = maintenance.mean_dist2_to_3nn(pts.to(DEV)).cpu()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-9)


def test_knn3_rejects_too_few_points():
    from vings_mono_amd._lib import VingsError
    with pytest.raises(VingsError):
        maintenance.mean_dist2_to_3nn(torch.rand(3, 3, device=DEV))
    assert maintenance.mean_dist2_to_3nn(torch.zeros(0, 3, device=DEV)).shape == (0,)


def test_add_splats_extends_parameters_state_and_records():
    dev = torch.device(DEV)
    W, H = 160, 96
    raw = synth.make_splats(3000, W, H, n_views=2, seed=21)
    extra = synth.make_splats(700, W, H, n_views=2, seed=22)
    intr = synth.intrinsic_for(W, H)
    w2c = torch.linalg.inv(synth.window_poses(2)[1]).to(dev)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    with torch.no_grad():
        pr = harness.render(gtp, w2c, intr)
        gt = {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(),
              "depth_cov": torch.ones_like(pr["depth"])}
    m = harness.SurfelMapper(raw, dev)
    cam = harness.view_settings(w2c, intr)
    m.fused_view(cam, intr, gt, w2c[:3, :3].cpu(), 0)
    before = {k: m.params[k].detach().clone() for k in m.NAMES}
    mom = {k: m.optimizer.state[m.params[k]]["exp_avg"].clone() for k in m.NAMES}
    new = maintenance.new_splat_params(extra["_xyz"].to(dev), extra["_rgb"].to(dev),
                                       extra["_rotation"].to(dev))
    # the reference's initial values: isotropic log-scale from the 3-NN distance, opacity 0.8
    d2 = maintenance.mean_dist2_to_3nn(extra["_xyz"].to(dev)).clamp_min(1e-7)
    assert torch.equal(new["_scaling"], torch.log(torch.sqrt(d2))[:, None].repeat(1, 2))
    assert torch.allclose(torch.sigmoid(new["_opacity"]), torch.full((700, 1), 0.8, device=dev))
    total = maintenance.add_splats(m, new)
    assert total == 3700 == m.n
    for k in m.NAMES:
        assert torch.equal(m.params[k].detach()[:3000], before[k])
        assert torch.equal(m.params[k].detach()[3000:], new[k].reshape(700, -1))
        st = m.optimizer.state[m.params[k]]
        assert torch.equal(st["exp_avg"][:3000], mom[k])
        assert float(st["exp_avg"][3000:].abs().max()) == 0.0
        assert float(st["exp_avg_sq"][3000:].abs().max()) == 0.0
    assert m.global_scores.shape == (3700, 2) and m.stable_mask.shape == (3700,)
    assert float(m.global_scores[3000:].abs().max()) == 0.0
    loss = m.fused_view(cam, intr, gt, w2c[:3, :3].cpu(), 1)    # the grown map still trains
    assert torch.isfinite(loss).all()


def _gold():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                "harness_golden.npz"))


def test_keyframe_pointcloud_matches_reference_golden():
    """Against the reference's get_pointcloud run on CPU with the same permutation; float
    tolerance 1e-5 relative + 1e-6 absolute (its matrix product may sum in another order)."""
    gold = _gold()
    fu, fv, cu, cv = [float(v) for v in gold["pc_intr"]]
    intr = {"fu": fu, "fv": fv, "cu": cu, "cv": cv, "H": 36, "W": 52}
    t = lambda k: torch.from_numpy(gold[k]).to(DEV)
    xyz, col = maintenance.keyframe_pointcloud(
        torch.from_numpy(gold["pc_c2w"]), intr, t("pc_rgb"), t("pc_depth"), t("pc_accum"), 900, 0.6,
        perm=torch.from_numpy(gold["pc_perm"]))
    assert xyz.shape == gold["pc_xyz"].shape                    # sample count follows the accum share
    np.testing.assert_allclose(xyz.cpu().numpy(), gold["pc_xyz"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(col.cpu().numpy(), gold["pc_col"])
    # first frame: no opacity map, every valid pixel counts
    xyz0, _ = maintenance.keyframe_pointcloud(
        torch.from_numpy(gold["pc_c2w"]), intr, t("pc_rgb"), t("pc_depth"), None, 900, 0.6)
    assert xyz0.shape[0] == 900


def test_depth_refresh_matches_reference_expression():
    H, W = 96, 160
    g = torch.Generator().manual_seed(5)
    dsum = torch.rand(H, W, generator=g) * 4
    alpha = torch.rand(H, W, generator=g)
    alpha[:10] = 0.0                      # 0/0 -> nan and x/0 -> inf: both count as no depth
    dsum[:5] = 0.0
    dsum[40:50, 30:60] = 0.0              # rendered depth 0: the old value stays
    dense = torch.rand(H, W, generator=g) + 0.1
    # loop_rectify.py:420-426 with gaussian_base.py:205-209
    depth = torch.nan_to_num(dsum / alpha, 0, 0)
    want_dense = dense.clone()
    want_dense[depth > 0] = depth[depth > 0]
    want_coarse = want_dense.reshape(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3)[..., 3:5, 3:5] \
        .reshape(H // 8, W // 8, -1).mean(dim=-1)
    from vings_mono_amd import _lib
    L = _lib.lib()
    d_dense = dense.to(DEV)
    d_coarse = torch.empty(H // 8, W // 8, device=DEV)
    a, b = dsum.to(DEV), alpha.to(DEV)
    _lib.check(L.vings_depth_refresh(H, W, _lib.ptr(a), _lib.ptr(b), _lib.ptr(d_dense),
                                     _lib.ptr(d_coarse), _lib.stream_ptr(torch.device(DEV))), "refresh")
    assert torch.equal(d_dense.cpu(), want_dense)
    torch.testing.assert_close(d_coarse.cpu(), want_coarse, rtol=1e-6, atol=0)
    with pytest.raises(_lib.VingsError):
        _lib.check(L.vings_depth_refresh(H - 1, W, _lib.ptr(a), _lib.ptr(b), _lib.ptr(d_dense),
                                         _lib.ptr(d_coarse), _lib.stream_ptr(torch.device(DEV))), "refresh")


def _frame_and_mapper(n=5000, seed=31):
    dev = torch.device(DEV)
    W, H = 160, 96
    raw = synth.make_splats(n, W, H, n_views=2, seed=seed)
    intr = synth.intrinsic_for(W, H)
    c2w = synth.window_poses(2)[1].to(dev)
    w2c = torch.linalg.inv(c2w)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    with torch.no_grad():
        pr = harness.render(gtp, w2c, intr)
    return raw, intr, c2w, w2c, pr


def test_add_new_frame_prunes_large_and_adds_points():
    dev = torch.device(DEV)
    raw, intr, c2w, w2c, pr = _frame_and_mapper()
    H, W = pr["depth"].shape[-2:]
    m = harness.SurfelMapper(raw, dev)
    gt = {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(), "depth_cov": torch.ones_like(pr["depth"])}
    m.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), 0)
    with torch.no_grad():
        radii = harness.render({k: v.detach() for k, v in m.params.items()}, w2c, intr)["radii"]
    n_big = int((radii > 25).sum())
    gdepth = pr["depth"].clone()
    gdepth[:, :, :20] = 0.0                                  # no depth there: no new points
    frame = {"pose": c2w, "depth": gdepth.permute(1, 2, 0), "image": pr["rgb"].permute(1, 2, 0),
             "intrinsic": intr}
    n0 = m.n
    pruned, added = maintenance.add_new_frame(m, frame, accum_thresh=0.5, n_points=2000)
    assert pruned == n_big
    assert 0 < added <= 2000 and m.n == n0 - pruned + added
    assert len(m.optimizer.state) == 0                        # fresh optimiser, as the reference builds
    new_xyz = m.params["_xyz"].detach()[-added:]
    # every new point lies on the frame's depth surface: it projects back inside the valid columns
    cam = (w2c[:3, :3] @ new_xyz.T + w2c[:3, 3:4]).T
    v = cam[:, 0] / cam[:, 2] * intr["fv"] + intr["cv"]
    u = cam[:, 1] / cam[:, 2] * intr["fu"] + intr["cu"]
    assert float(v.min()) >= 19.5 and float(v.max()) <= W - 0.5 and float(u.min()) >= -0.5
    got_depth = gdepth[0][u.round().long().clamp(0, H - 1), v.round().long().clamp(0, W - 1)]
    torch.testing.assert_close(cam[:, 2], got_depth, rtol=1e-4, atol=1e-5)
    assert torch.allclose(torch.sigmoid(m.params["_opacity"].detach()[-added:]),
                          torch.full((added, 1), 0.8, device=dev))
    loss = m.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), 1)
    assert torch.isfinite(loss).all()


def test_rectify_tracker_renders_depth_into_tracker_maps():
    dev = torch.device(DEV)
    raw, intr, c2w, w2c, pr = _frame_and_mapper(seed=33)
    H, W = pr["depth"].shape[-2:]
    m = harness.SurfelMapper(raw, dev)
    poses = [p.to(dev) for p in synth.window_poses(2)]
    dense = torch.full((2, H, W), 0.25, device=dev)
    coarse = torch.zeros((2, H // 8, W // 8), device=dev)
    maintenance.rectify_tracker(m, poses, intr, dense, coarse)
    for k, p in enumerate(poses):
        with torch.no_grad():
            d = harness.render({kk: v.detach() for kk, v in m.params.items()}, torch.linalg.inv(p), intr)["depth"][0]
        want = torch.where(d > 0, d, torch.full_like(d, 0.25))
        assert torch.equal(dense[k], want)
        wc = want.reshape(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3)[..., 3:5, 3:5].reshape(H // 8, W // 8, -1).mean(-1)
        torch.testing.assert_close(coarse[k], wc, rtol=1e-6, atol=0)
