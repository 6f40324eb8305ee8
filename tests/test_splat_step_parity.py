"""Per-splat stages of the view step (csrc/splat_step.hip through the C ABI) against the
numpy oracle (oracle/mapper_oracle.py), and the fused view step against the autograd-driven
formulation of the same step.  Tolerance: 1e-6 relative for the elementwise stages (float32
on both sides, libm-level differences in exp); bookkeeping integers and masks bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import mapper_oracle as mo
from vings_mono_amd import _lib, harness, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _call_finish(n, t, bookkeeping, this_kf, stable):
    L = _lib.lib()
    p = _lib.ptr
    out = {k: torch.empty_like(t[k2]) for k, k2 in (("_xyz", "d_means3D"), ("_rgb", "d_colors"),
                                                     ("_opacity", "d_opacities"),
                                                     ("_scaling", "d_scales"),
                                                     ("_rotation", "d_rotations"))}
    vis = torch.empty(n, dtype=torch.uint8, device=DEV)
    _lib.check(L.vings_splat_grad_finish(
        n, p(t["opac"]), p(t["scales"]), p(t["raw_rot"]), p(t["d_means3D"]), p(t["d_colors"]),
        p(t["d_opacities"]), p(t["d_scales"]), p(t["d_rotations"]), p(t["cur"]), p(t["radii"]),
        p(stable), p(t["glob"]), p(t["loc"]), p(t["kf_max"]), p(t["kf_id"]), this_kf, bookkeeping,
        p(out["_xyz"]), p(out["_rgb"]), p(out["_opacity"]), p(out["_scaling"]), p(out["_rotation"]),
        p(vis), _lib.stream_ptr(torch.device(DEV))), "vings_splat_grad_finish")
    torch.cuda.synchronize()
    return out, vis


def _host_inputs(n, seed):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.normal(size=s).astype(np.float32)
    h = {"raw_opac": f(n, 1) * 2, "raw_scal": f(n, 2) * 0.5 - 2.0, "raw_rot": f(n, 4),
         "d_means3D": f(n, 3), "d_colors": f(n, 3), "d_opacities": f(n, 1), "d_scales": f(n, 2),
         "d_rotations": f(n, 4),
         "cur": np.abs(f(n, 2)) * (rng.random((n, 1)) > 0.3),      # some splats unseen: s0 = 0
         "radii": (rng.integers(-1, 30, n) * (rng.random(n) > 0.2)).astype(np.int32),
         "glob": np.abs(f(n, 2)) * 50, "loc": np.abs(f(n, 2)), "kf_max": np.abs(f(n)),
         "kf_id": rng.integers(0, 9, n).astype(np.int64), "stable": rng.random(n) > 0.8}
    h["glob"][: n // 10, 0] = 9999.9                                  # exercises the 1e4 clamp
    return h


@pytest.mark.parametrize("n", [1, 255, 4097])
def test_activate_and_grad_finish_parity(n):
    h = _host_inputs(n, 100 + n)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in h.items()}
    L = _lib.lib()
    p = _lib.ptr
    opac, scales, rot = (torch.empty(n, 1, device=DEV), torch.empty(n, 2, device=DEV),
                         torch.empty(n, 4, device=DEV))
    _lib.check(L.vings_splat_activate(n, p(t["raw_opac"]), p(t["raw_scal"]), p(t["raw_rot"]), p(opac),
                                      p(scales), p(rot), _lib.stream_ptr(torch.device(DEV))),
               "vings_splat_activate")
    o_o, o_s, o_q = mo.activate(h["raw_opac"], h["raw_scal"], h["raw_rot"])
    np.testing.assert_allclose(opac.cpu().numpy(), o_o, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(scales.cpu().numpy(), o_s, rtol=1e-6)
    np.testing.assert_allclose(rot.cpu().numpy(), o_q, rtol=1e-6, atol=1e-7)

    t.update(opac=opac, scales=scales)
    d = {"means3D": h["d_means3D"], "colors": h["d_colors"], "opacities": h["d_opacities"],
         "scales": h["d_scales"], "rotations": h["d_rotations"]}
    want, w_vis, w_glob, w_loc, w_kfmax, w_kfid = mo.grad_finish(
        opac.cpu().numpy(), scales.cpu().numpy(), h["raw_rot"], d, h["cur"], h["radii"], h["stable"],
        h["glob"], h["loc"], h["kf_max"], h["kf_id"], 7)
    got, vis = _call_finish(n, t, 1, 7, t["stable"].view(torch.uint8))
    for k in want:
        # the rotation gradient is a difference g - n (n.g): absolute floor of 1e-6 of the array
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k], rtol=2e-6,
                                   atol=1e-6 * float(np.abs(want[k]).max()) + 1e-9, err_msg=k)
    assert np.array_equal(vis.cpu().numpy().astype(bool), w_vis)
    np.testing.assert_allclose(t["glob"].cpu().numpy(), w_glob, rtol=1e-7)
    np.testing.assert_allclose(t["loc"].cpu().numpy(), w_loc, rtol=1e-7)
    np.testing.assert_array_equal(t["kf_max"].cpu().numpy(), w_kfmax)
    np.testing.assert_array_equal(t["kf_id"].cpu().numpy(), w_kfid)


def test_grad_finish_without_bookkeeping_leaves_scores_alone():
    n = 300
    h = _host_inputs(n, 9)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in h.items()}
    o, s, _ = mo.activate(h["raw_opac"], h["raw_scal"], h["raw_rot"])
    t.update(opac=torch.from_numpy(o).to(DEV), scales=torch.from_numpy(s).to(DEV))
    before = {k: t[k].clone() for k in ("glob", "loc", "kf_max", "kf_id")}
    got, vis = _call_finish(n, t, 0, 3, None)           # NULL stable mask: nothing is stable
    for k in before:
        assert torch.equal(before[k], t[k]), k
    assert np.array_equal(vis.cpu().numpy().astype(bool), h["radii"] > 0)
    d = {"means3D": h["d_means3D"], "colors": h["d_colors"], "opacities": h["d_opacities"],
         "scales": h["d_scales"], "rotations": h["d_rotations"]}
    want = mo.grad_finish(o, s, h["raw_rot"], d, h["cur"], h["radii"], np.zeros(n, bool), h["glob"],
                          h["loc"], h["kf_max"], h["kf_id"], 3)[0]
    for k in want:
        # the rotation gradient is a difference g - n (n.g): absolute floor of 1e-6 of the array
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k], rtol=2e-6,
                                   atol=1e-6 * float(np.abs(want[k]).max()) + 1e-9, err_msg=k)


def test_empty_input_is_a_no_op():
    L = _lib.lib()
    assert L.vings_splat_activate(0, None, None, None, None, None, None, None) == 0


def test_fused_view_matches_autograd_view():
    """Two mapper replicas take three view steps, one through torch autograd + PyTorch
    bookkeeping, one through the fused C-ABI sequence; parameters, Adam state and scores
    agree to 2e-3 of each array's largest entry (float32 gradient sums in different order,
    amplified by Adam's normalisation of tiny gradients)."""
    dev = torch.device(DEV)
    W, H = 160, 96
    raw = synth.make_splats(5000, W, H, n_views=3, seed=21)
    intr = synth.intrinsic_for(W, H)
    c2w = synth.window_poses(3)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    views = []
    with torch.no_grad():
        for k in range(3):
            w2c = torch.linalg.inv(c2w[k]).to(dev)
            pr = harness.render(gtp, w2c, intr)
            views.append((w2c, {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(),
                                "depth_cov": torch.ones_like(pr["depth"])}))
    a, b = harness.SurfelMapper(raw, dev), harness.SurfelMapper(raw, dev)
    a.stable_mask[::7] = True
    b.stable_mask[::7] = True
    for k, (w2c, gt) in enumerate(views):
        radii, cur = a.backward_view(w2c, intr, gt)
        a.finish_view(radii, cur, k)
        b.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), k)
    torch.cuda.synchronize()
    assert abs(float(a.last_loss) - float(b.last_loss)) <= 1e-4 * abs(float(a.last_loss))
    # the best-keyframe record may flip only where two views score a splat within rounding
    assert float((a.globalkf_id == b.globalkf_id).float().mean()) >= 0.999

    def close(x, y, name, tol=2e-3):
        scale = float(x.abs().max())
        err = float((x - y).abs().max())
        assert err <= tol * scale + 1e-12, f"{name}: {err:.3e} vs {scale:.3e}"

    for k in a.NAMES:
        close(a.params[k].detach() - raw[k].to(dev), b.params[k].detach() - raw[k].to(dev),
              "update of " + k, tol=2e-2)
        close(a.optimizer.state[a.params[k]]["exp_avg"], b.optimizer.state[b.params[k]]["exp_avg"],
              "exp_avg " + k)
    close(a.global_scores, b.global_scores, "global_scores", tol=1e-4)
    close(a.local_scores, b.local_scores, "local_scores", tol=1e-4)
    close(a.globalkf_max_scores, b.globalkf_max_scores, "kf_max", tol=1e-4)


def test_fused_view_with_half_the_map_stable():
    """Half of the splats under the stable mask: the view step that skips their gradients
    (stable_skip) against the one that computes and then ignores them.  Same parameters, Adam
    moments and score records (to what two runs of the float-atomic backward differ by); the
    stable splats themselves do not move at all."""
    dev = torch.device(DEV)
    W, H = 160, 96
    raw = synth.make_splats(5000, W, H, n_views=3, seed=23)
    intr = synth.intrinsic_for(W, H)
    c2w = synth.window_poses(3)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    a, b = harness.SurfelMapper(raw, dev), harness.SurfelMapper(raw, dev)
    flagged = torch.rand(5000, generator=torch.Generator().manual_seed(4)).to(dev) < 0.5
    for m in (a, b):
        m.stable_mask[:] = flagged
    assert b.note_stable_mask() and not a.stable_skip
    for k in range(3):
        w2c = torch.linalg.inv(c2w[k]).to(dev)
        with torch.no_grad():
            pr = harness.render(gtp, w2c, intr)
        gt = {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(), "depth_cov": torch.ones_like(pr["depth"])}
        for m in (a, b):
            m.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), k)
    torch.cuda.synchronize()

    def close(x, y, name, tol):
        scale = float(x.abs().max())
        assert float((x - y).abs().max()) <= tol * scale + 1e-12, name

    for k in a.NAMES:
        assert torch.equal(b.params[k].detach()[flagged], raw[k].to(dev)[flagged]), k
        assert float((a.params[k].detach() - raw[k].to(dev)).abs().max()) > 0
        close(a.params[k].detach() - raw[k].to(dev), b.params[k].detach() - raw[k].to(dev), k, 2e-2)
        close(a.optimizer.state[a.params[k]]["exp_avg"], b.optimizer.state[b.params[k]]["exp_avg"], k, 2e-3)
    close(a.global_scores, b.global_scores, "global_scores", 1e-5)
    close(a.local_scores, b.local_scores, "local_scores", 1e-5)
    close(a.globalkf_max_scores, b.globalkf_max_scores, "kf_max", 1e-5)
    assert float(b.local_scores[flagged].abs().max()) > 0
    assert float((a.globalkf_id == b.globalkf_id).float().mean()) >= 0.999


@pytest.mark.parametrize("bookkeeping", [1, 0])
def test_project_backward_finish_equals_the_two_calls(bookkeeping):
    """vings_surfel_project_backward_finish against vings_surfel_project_backward followed by
    vings_splat_grad_finish on the same blend-pass output: every output bit-identical, except
    that the fused call leaves zeros where the two calls leave the (never read) gradients of
    stable splats."""
    from vings_mono_amd import surfel
    dev = torch.device(DEV)
    L = _lib.lib()
    p = _lib.ptr
    W, H, n = 160, 96, 4000
    raw = {k: v.to(dev) for k, v in synth.make_splats(n, W, H, n_views=2, seed=7).items()}
    intr = synth.intrinsic_for(W, H)
    w2c = torch.linalg.inv(synth.window_poses(2)[1]).to(dev)
    act = synth.activate(raw)
    opac, scales, rot = (act[k].contiguous() for k in ("opacities", "scales", "rotations"))
    xyz, rgb = raw["_xyz"].contiguous(), raw["_rgb"].contiguous()
    st, aux = surfel.rasterize_forward(xyz, rgb, opac, scales, rot, harness.view_settings(w2c, intr))
    g = torch.Generator(device="cpu").manual_seed(3)
    g_color = torch.randn(3, H, W, generator=g).to(dev)
    g_allmap = torch.randn(surfel.AUX_CHANNELS, H, W, generator=g).to(dev)
    dgeom = surfel.blend_backward(st, aux, g_color, g_allmap)
    stable = (torch.arange(n, device=dev) % 5 == 0).to(torch.uint8)
    stream = _lib.stream_ptr(dev)

    def records():
        gen = torch.Generator(device="cpu").manual_seed(11)
        return [torch.rand(n, 2, generator=gen).to(dev) * 3, torch.rand(n, 2, generator=gen).to(dev),
                torch.rand(n, generator=gen).to(dev) * 0.05, torch.zeros(n, dtype=torch.int64, device=dev)]

    def outputs():
        return [torch.full((n, w), 7.0, device=dev) for w in (3, 3, 1, 2, 4)] + \
               [torch.full((n,), 9, dtype=torch.uint8, device=dev)]

    # two calls
    d = [torch.empty((n, w), device=dev) for w in (3, 2, 4, 1, 3)]     # xyz, scales, rot, opac, colour
    cur = torch.empty((n, 2), device=dev)
    _lib.check(L.vings_surfel_project_backward(
        n, H, W, 1.0, p(xyz), p(scales), p(rot), p(aux[0]), p(aux[1]), p(st.radii), p(dgeom),
        p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(d[4]), None, p(cur), stream), "project_backward")
    rec_a, out_a = records(), outputs()
    _lib.check(L.vings_splat_grad_finish(
        n, p(opac), p(scales), p(raw["_rotation"]), p(d[0]), p(d[4]), p(d[3]), p(d[1]), p(d[2]),
        p(cur), p(st.radii), p(stable), p(rec_a[0]), p(rec_a[1]), p(rec_a[2]), p(rec_a[3]), 5,
        bookkeeping, *[p(t) for t in out_a], stream), "grad_finish")
    # one call
    rec_b, out_b = records(), outputs()
    cur_b = torch.empty((n, 2), device=dev)
    _lib.check(L.vings_surfel_project_backward_finish(
        n, H, W, 1.0, p(xyz), p(scales), p(rot), p(aux[0]), p(aux[1]), p(st.radii), p(dgeom),
        p(opac), p(raw["_rotation"]), p(stable), p(rec_b[0]), p(rec_b[1]), p(rec_b[2]), p(rec_b[3]),
        5, bookkeeping, *[p(t) for t in out_b], p(cur_b), None, None, stream), "project_backward_finish")
    torch.cuda.synchronize()
    assert float(out_a[0].abs().max()) > 0 and int(out_a[5].sum()) > 0
    flagged = stable.bool()
    for x, y in zip(out_a[:5], out_b[:5]):
        assert torch.equal(x[~flagged], y[~flagged])
        assert float(x[flagged].abs().max()) > 0 and float(y[flagged].abs().max()) == 0
    for x, y in zip(out_a[5:] + rec_a + [cur], out_b[5:] + rec_b + [cur_b]):
        assert torch.equal(x, y)
    # scores_grad is optional
    _lib.check(L.vings_surfel_project_backward_finish(
        n, H, W, 1.0, p(xyz), p(scales), p(rot), p(aux[0]), p(aux[1]), p(st.radii), p(dgeom),
        p(opac), p(raw["_rotation"]), None, p(rec_b[0]), p(rec_b[1]), p(rec_b[2]), p(rec_b[3]),
        6, 0, *[p(t) for t in out_b], None, None, None, stream), "project_backward_finish")
    torch.cuda.synchronize()
    if bookkeeping == 0:
        # multi-GPU form: straight into the exchange buffer == two calls + vings_splat_pack_exchange
        cols = L.vings_splat_pack_columns()
        keys0 = torch.randint(0, 1 << 40, (n, 2), generator=torch.Generator().manual_seed(2)).to(dev)
        pk_a, keys_a = torch.empty((cols, n), device=dev), keys0.clone()
        _lib.check(L.vings_splat_pack_exchange(n, *[p(t) for t in out_a], p(cur), 5, p(pk_a), p(keys_a),
                                               stream), "pack_exchange")
        pk_b, keys_b = torch.empty((cols, n), device=dev), keys0.clone()
        _lib.check(L.vings_surfel_project_backward_finish(
            n, H, W, 1.0, p(xyz), p(scales), p(rot), p(aux[0]), p(aux[1]), p(st.radii), p(dgeom),
            p(opac), p(raw["_rotation"]), p(stable), p(rec_b[0]), None, None, None, 5, 0,
            None, None, None, None, None, None, None, p(pk_b), p(keys_b), stream),
            "project_backward_finish (packed)")
        torch.cuda.synchronize()
        assert torch.equal(pk_a[:, ~flagged], pk_b[:, ~flagged]) and torch.equal(keys_a, keys_b)
        assert torch.equal(pk_a[13:, flagged], pk_b[13:, flagged])        # score and visibility columns
        assert float(pk_b[:13, flagged].abs().max()) == 0
        assert not torch.equal(keys_a, keys0)


@pytest.mark.parametrize("n_chunks,extra_touched", [(1, 0), (3, 0), (4, 60)])
def test_compact_exchange_of_one_rank_equals_the_single_gpu_step(n_chunks, extra_touched):
    """The compact multi-GPU tail (touched mask -> slot scan -> per-splat backward packed into
    chunks -> per chunk: score records + Adam; record keys merged afterwards) fed with ONE
    rank's blend-pass output, against the single-GPU tail on the same output
    (vings_surfel_project_backward_finish with bookkeeping + SparseGaussianAdam.step):
    parameters, both Adam moments and every score record bit-identical.  `extra_touched` rows
    that this rank does not see are added to the union, as another rank's view would."""
    from vings_mono_amd import parallel, surfel
    dev = torch.device(DEV)
    L = _lib.lib()
    p = _lib.ptr
    W, H, n = 160, 96, 4000
    raw = {k: v.to(dev).contiguous() for k, v in synth.make_splats(n, W, H, n_views=2, seed=7).items()}
    intr = synth.intrinsic_for(W, H)
    w2c = torch.linalg.inv(synth.window_poses(2)[1]).to(dev)
    act = synth.activate(raw)
    opac, scales, rot = (act[k].contiguous() for k in ("opacities", "scales", "rotations"))
    xyz, rgb = raw["_xyz"], raw["_rgb"]
    st, aux = surfel.rasterize_forward(xyz, rgb, opac, scales, rot, harness.view_settings(w2c, intr))
    g = torch.Generator(device="cpu").manual_seed(3)
    g_color = torch.randn(3, H, W, generator=g).to(dev)
    g_allmap = torch.randn(surfel.AUX_CHANNELS, H, W, generator=g).to(dev)
    dgeom = surfel.blend_backward(st, aux, g_color, g_allmap)
    stable = (torch.arange(n, device=dev) % 5 == 0)
    stream = _lib.stream_ptr(dev)
    lrs = {k + "_lr": v for k, v in zip(harness.SurfelMapper.NAMES, (1e-3, 2e-3, 3e-2, 4e-3, 1e-3))}

    def mapper():
        m = harness.SurfelMapper({k: v.clone() for k, v in raw.items()}, dev, lr=lrs)
        gen = torch.Generator(device="cpu").manual_seed(11)
        m.global_scores = (torch.rand(n, 2, generator=gen) * 3).to(dev)
        m.local_scores = torch.rand(n, 2, generator=gen).to(dev)
        m.globalkf_max_scores = (torch.rand(n, generator=gen) * 0.05).to(dev)
        m.stable_mask = stable.clone()
        for k in m.NAMES:                                    # moments of an earlier step
            stt = m.optimizer._moments(m.params[k])
            stt["exp_avg"].copy_(torch.randn(m.params[k].shape, generator=gen) * 1e-3)
            stt["exp_avg_sq"].copy_(torch.rand(m.params[k].shape, generator=gen) * 1e-6)
        return m

    # single-GPU tail
    a = mapper()
    ga = [torch.zeros_like(a.params[k]) for k in a.NAMES]
    for k, t in zip(a.NAMES, ga):
        a.params[k].grad = t
    vis = torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.check(L.vings_surfel_project_backward_finish(
        n, H, W, 1.0, p(xyz), p(scales), p(rot), p(aux[0]), p(aux[1]), p(st.radii), p(dgeom),
        p(opac), p(raw["_rotation"]), p(a.stable_mask.view(torch.uint8)), p(a.global_scores),
        p(a.local_scores), p(a.globalkf_max_scores), p(a.globalkf_id), 5, 1, *[p(t) for t in ga],
        p(vis), None, None, None, stream), "project_backward_finish")
    unseen = torch.nonzero(st.radii <= 0).reshape(-1)[:extra_touched]
    assert unseen.numel() == extra_touched
    # a splat another rank sees takes the Adam step here too (with that rank's gradient; none
    # in this test), unless it is stable
    vis[unseen] = (~stable[unseen]).to(torch.uint8)
    a.optimizer.step(vis, n)

    # compact tail, the "all-reduce" being the identity
    b = mapper()
    touched = torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.check(L.vings_splat_touched(n, p(st.radii), p(touched), stream), "touched")
    touched[unseen] = 1
    slot = torch.empty(n, dtype=torch.int32, device=dev)
    rows = torch.full((n,), -7, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    ws_bytes = int(L.vings_splat_compact_workspace_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(L.vings_splat_compact_plan_device(n, p(touched), p(slot), p(ws), ws_bytes, p(count),
                                                 stream), "plan_device")
    m = int(count.item())
    assert m == int(touched.sum()) and m >= int((st.radii > 0).sum())
    chunk_rows, used = parallel.chunk_layout(m, n_chunks)
    assert used == -(-m // chunk_rows) and used <= n_chunks and chunk_rows % 256 == 0
    cols = L.vings_splat_exchange_columns()
    packed = torch.full((cols * chunk_rows * used,), float("nan"), device=dev)
    keys = torch.zeros((n, 2), dtype=torch.int64, device=dev)
    _lib.check(L.vings_surfel_project_backward_pack(
        n, H, W, 1.0, p(xyz), p(scales), p(rot), p(aux[0]), p(aux[1]), p(st.radii), p(dgeom),
        p(opac), p(raw["_rotation"]), p(b.global_scores), 5, None, p(packed), p(keys), p(slot),
        p(rows), chunk_rows, stream), "project_backward_pack")
    assert torch.equal(rows[:m].long(), torch.nonzero(touched).reshape(-1))
    PP, MM, VV, LR, b1, b2, eps = b.optimizer.row_step_args(n, (3, 3, 1, 2, 4))
    for k in range(used):
        chunk = packed[k * cols * chunk_rows:(k + 1) * cols * chunk_rows]
        first, cnt = k * chunk_rows, min(chunk_rows, m - k * chunk_rows)
        _lib.check(L.vings_splat_exchange_step(
            cnt, chunk_rows, p(chunk), rows.data_ptr() + 4 * first, PP, MM, VV, LR, b1, b2, eps,
            p(b.stable_mask.view(torch.uint8)), p(b.global_scores), p(b.local_scores), stream),
            "exchange_step")
    _lib.check(L.vings_splat_merge_records(n, p(keys), p(b.local_scores), p(b.globalkf_max_scores),
                                           p(b.globalkf_id), stream), "merge_records")
    torch.cuda.synchronize()
    assert int(vis.sum()) > 0
    for k in a.NAMES:
        assert not torch.equal(a.params[k].detach(), raw[k]), k
        assert torch.equal(a.params[k].detach(), b.params[k].detach()), k
        for mom in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(a.optimizer.state[a.params[k]][mom],
                               b.optimizer.state[b.params[k]][mom]), (k, mom)
    assert torch.equal(a.global_scores, b.global_scores)
    assert torch.equal(a.local_scores, b.local_scores)
    assert torch.equal(a.globalkf_max_scores, b.globalkf_max_scores)
    assert torch.equal(a.globalkf_id, b.globalkf_id)


def test_projection_with_inline_activation_equals_activate_then_project():
    """rasterize_forward(raw=...) (activations applied inside the projection kernel) against
    vings_splat_activate followed by the plain forward pass: activated arrays and every
    forward output bit-identical, on the first (exact) and the second (speculative) launch."""
    from vings_mono_amd import surfel
    dev = torch.device(DEV)
    L = _lib.lib()
    p = _lib.ptr
    W, H, n = 176, 112, 6000
    raw = {k: v.to(dev).contiguous() for k, v in synth.make_splats(n, W, H, n_views=2, seed=9).items()}
    raw["_rotation"][5] = 0.0                                  # degenerate quaternion
    intr = synth.intrinsic_for(W, H)
    settings = harness.view_settings(torch.linalg.inv(synth.window_poses(2)[1]).to(dev), intr)
    e = lambda *s: torch.empty(*s, device=dev)
    o1, s1, r1 = e(n, 1), e(n, 2), e(n, 4)
    _lib.check(L.vings_splat_activate(n, p(raw["_opacity"]), p(raw["_scaling"]), p(raw["_rotation"]),
                                      p(o1), p(s1), p(r1), _lib.stream_ptr(dev)), "activate")
    want = surfel.rasterize_forward(raw["_xyz"], raw["_rgb"], o1, s1, r1, settings, keep_keys=True)[0]
    surfel._D_HINT.clear()
    for _ in range(2):
        o2, s2, r2 = (torch.full_like(t, 5.0) for t in (o1, s1, r1))
        dg = torch.full((n, surfel.GRAD_STRIDE), 5.0, device=dev)
        got = surfel.rasterize_forward(raw["_xyz"], raw["_rgb"], o2, s2, r2, settings, keep_keys=True,
                                       raw=(raw["_opacity"], raw["_scaling"], raw["_rotation"]),
                                       dgeom=dg)[0]
        # the rows the backward pass adds into are cleared, the others left alone
        seen = got.radii > 0
        assert bool((dg[seen] == 0).all()) and bool((dg[~seen] == 5.0).all()) and int((~seen).sum()) > 0
        assert torch.equal(o1, o2) and torch.equal(s1, s2) and torch.equal(r1, r2)
        assert got.D == want.D and want.D > 0
        for name in ("geom", "depth", "rect", "ranges", "radii", "n_contrib", "color", "allmap", "final_T"):
            a, b = getattr(want, name), getattr(got, name)
            if name == "geom":                                  # rows of culled splats are not written
                vis = want.radii > 0
                a, b = a[vis], b[vis]
            assert torch.equal(a, b), name
        assert torch.equal(want.keys[:want.D], got.keys[:got.D])


def test_fused_view_with_nothing_in_view_changes_nothing():
    """A view step whose camera looks away from every splat (no instance, empty tile lists, the
    ordered blend passes walking 0-length lists): finite loss, parameters, Adam moments and
    score records untouched, and the next ordinary step still works (eager and as a graph
    candidate alike: the empty step must not poison the instance-count hint)."""
    dev = torch.device(DEV)
    W, H = 160, 96
    raw = synth.make_splats(3000, W, H, n_views=2, seed=27)
    intr = synth.intrinsic_for(W, H)
    c2w = synth.window_poses(2)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    w2c = torch.linalg.inv(c2w[0]).to(dev)
    with torch.no_grad():
        pr = harness.render(gtp, w2c, intr)
    gt = {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(), "depth_cov": torch.ones_like(pr["depth"])}
    m = harness.SurfelMapper(raw, dev)
    m.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), 0)   # an ordinary step
    torch.cuda.synchronize()
    before = {k: m.params[k].detach().clone() for k in m.NAMES}
    mom = {k: m.optimizer.state[m.params[k]]["exp_avg"].clone() for k in m.NAMES}
    rec = (m.global_scores.clone(), m.local_scores.clone(), m.globalkf_id.clone())
    away = torch.diag(torch.tensor([-1.0, 1.0, -1.0, 1.0], device=dev)) @ w2c
    vec = m.fused_view(harness.view_settings(away, intr), intr, gt, away[:3, :3].cpu(), 1)
    torch.cuda.synchronize()
    assert torch.isfinite(vec).all() and m.last_stats()["D"] == 0
    for k in m.NAMES:
        assert torch.equal(m.params[k].detach(), before[k]), k
        assert torch.equal(m.optimizer.state[m.params[k]]["exp_avg"], mom[k]), k
    assert torch.equal(m.global_scores, rec[0]) and torch.equal(m.local_scores, rec[1])
    assert torch.equal(m.globalkf_id, rec[2])
    m.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), 0)
    torch.cuda.synchronize()
    assert m.last_stats()["D"] > 0
    assert not torch.equal(m.params["_xyz"].detach(), before["_xyz"])


@pytest.mark.parametrize("W,H,n", [(20, 12, 60), (16, 16, 1), (47, 33, 400)])
def test_fused_view_on_tiny_images(W, H, n):
    """One tile, a partial tile, a handful of splats: the fused step (ordered blend passes, tile
    order of 1..6 tiles, batches of far fewer than 256 records) against the autograd-driven
    step on the same inputs."""
    dev = torch.device(DEV)
    raw = synth.make_splats(n, W, H, n_views=1, seed=29, z_range=(2.0, 6.0))
    intr = synth.intrinsic_for(W, H)
    w2c = torch.linalg.inv(synth.window_poses(1)[0]).to(dev)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    with torch.no_grad():
        pr = harness.render(gtp, w2c, intr)
    gt = {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(), "depth_cov": torch.ones_like(pr["depth"])}
    a, b = harness.SurfelMapper(raw, dev), harness.SurfelMapper(raw, dev)
    radii, cur = a.backward_view(w2c, intr, gt)
    a.finish_view(radii, cur, 0)
    b.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), 0)
    torch.cuda.synchronize()
    assert abs(float(a.last_loss) - float(b.last_loss)) <= 1e-4 * abs(float(a.last_loss)) + 1e-7
    for k in a.NAMES:
        x = a.optimizer.state[a.params[k]]["exp_avg"]
        y = b.optimizer.state[b.params[k]]["exp_avg"]
        assert torch.isfinite(y).all()
        assert float((x - y).abs().max()) <= 2e-3 * float(x.abs().max()) + 1e-12, k
