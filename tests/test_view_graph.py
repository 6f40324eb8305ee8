"""The mapper's view step replayed from a HIP graph (`SurfelMapper.fused_view_graphed`) against
the eager fused step: same map after several iterations over a window, and a replay whose
instance list outgrew its graph leaves the map untouched until the step is repeated."""
import pytest
import torch

from vings_mono_amd import harness, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _window(n, W, H, n_views, seed):
    dev = torch.device(DEV)
    raw = synth.make_splats(n, W, H, n_views=n_views, seed=seed)
    intr = synth.intrinsic_for(W, H)
    c2w = synth.window_poses(n_views)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    views = []
    with torch.no_grad():
        for k in range(n_views):
            w2c = torch.linalg.inv(c2w[k]).to(dev)
            pr = harness.render(gtp, w2c, intr)
            gt = {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(),
                  "depth_cov": torch.ones_like(pr["depth"])}
            views.append((harness.view_settings(w2c, intr), gt, w2c[:3, :3].cpu()))
    return raw, intr, views


def _close(a, b, raw):
    dev = torch.device(DEV)
    for k in a.NAMES:
        ua, ub = a.params[k].detach() - raw[k].to(dev), b.params[k].detach() - raw[k].to(dev)
        scale = float(ua.abs().max())
        assert scale > 0, k
        assert float((ua - ub).abs().max()) <= 2e-2 * scale, k
    for name in ("global_scores", "local_scores", "globalkf_max_scores"):
        x, y = getattr(a, name), getattr(b, name)
        assert float((x - y).abs().max()) <= 1e-4 * float(x.abs().max()), name
    assert float((a.globalkf_id == b.globalkf_id).float().mean()) >= 0.999


def test_graph_replay_equals_eager_view_steps():
    dev = torch.device(DEV)
    raw, intr, views = _window(6000, 320, 192, 3, seed=71)
    a, b = harness.SurfelMapper(raw, dev), harness.SurfelMapper(raw, dev)
    a.stable_mask[::7] = True
    b.stable_mask[::7] = True
    for it in range(4):                      # eager, capture + replay, replay, replay
        for k, (cam, gt, R) in enumerate(views):
            la = a.fused_view(cam, intr, gt, R, k)
            lb = b.fused_view_graphed(k, cam, intr, gt, R, k)
            if it == 0:
                assert b._view_graphs[k]["graph"] is None
            else:
                assert b._view_graphs[k]["graph"] is not None
    torch.cuda.synchronize()
    assert abs(float(la[0]) - float(lb[0])) <= 1e-3 * abs(float(la[0]))
    for k in range(3):
        ent = b._view_graphs[k]
        assert 0 < int(ent["count"][2]) <= ent["capacity"]
    _close(a, b, raw)


def test_overgrown_replay_is_dropped_and_repeated():
    dev = torch.device(DEV)
    raw, intr, views = _window(5000, 256, 160, 1, seed=72)
    cam, gt, R = views[0]
    a, b = harness.SurfelMapper(raw, dev), harness.SurfelMapper(raw, dev)
    for _ in range(3):
        a.fused_view(cam, intr, gt, R, 0)
    b.fused_view_graphed(0, cam, intr, gt, R, 0)                 # eager: learns D
    ent = b._view_graphs[0]
    true_d = ent["D"]
    assert true_d > 8000
    ent["D"] = 64                                                # the graph gets far too little room
    before = {k: b.params[k].detach().clone() for k in b.NAMES}
    scores_before = b.global_scores.clone()
    b.fused_view_graphed(0, cam, intr, gt, R, 0)                 # captured + replayed: dropped
    torch.cuda.synchronize()
    assert int(ent["count"][2]) > ent["capacity"]
    for k in b.NAMES:
        assert torch.equal(b.params[k].detach(), before[k]), k
    assert torch.equal(b.global_scores, scores_before)
    b.fused_view_graphed(0, cam, intr, gt, R, 0)                 # repeats the dropped step, re-captures
    torch.cuda.synchronize()
    assert ent["capacity"] > true_d // 2 and int(ent["count"][2]) <= ent["capacity"]
    _close(a, b, raw)


def test_new_addresses_recapture():
    dev = torch.device(DEV)
    raw, intr, views = _window(3000, 192, 128, 1, seed=73)
    cam, gt, R = views[0]
    b = harness.SurfelMapper(raw, dev)
    b.fused_view_graphed(0, cam, intr, gt, R, 0)
    b.fused_view_graphed(0, cam, intr, gt, R, 0)
    g0 = b._view_graphs[0]["graph"]
    assert g0 is not None
    gt2 = {k: gt[k].clone() for k in ("rgb", "depth", "depth_cov")}   # a new keyframe image
    b.fused_view_graphed(0, cam, intr, gt2, R, 0)                # eager again
    assert b._view_graphs[0]["graph"] is None
    b.fused_view_graphed(0, cam, intr, gt2, R, 0)
    assert b._view_graphs[0]["graph"] is not None and b._view_graphs[0]["graph"] is not g0
