"""Times the stages either side of the mapping hot path on one MI355X and prints one JSON
object: loop closure at the 5M-splat / 64-keyframe size (importance scoring, per-splat
re-anchor, tracker refresh), keyframe arrival (3-NN scale initialisation, back-projection,
pruning compaction), the tracker -> mapper packaging kernels (upsampling, depth
covariance, keyframe packaging) and one dense bundle-adjustment call on the reference's
own grid and on the 1080p/8 grid.  Every figure is the median of `--reps` timed calls between
device events on the stream the library launches on; algorithmic bytes are given beside the
time so the achieved bandwidth can be read off.

Usage:  python tools/bench_side_paths.py [--splats 5000000] [--keyframes 64] [--reps 5]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from vings_mono_amd import _lib, harness, maintenance, synth, tracker_package as TP   # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return float(np.median(out))


def entry(ms, nbytes=None, **extra):
    e = {"ms": round(ms, 4)}
    if nbytes is not None:
        e["algorithmic_MB"] = round(nbytes / 1e6, 2)
        e["GB_per_s"] = round(nbytes / ms / 1e6, 1)
    e.update(extra)
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splats", type=int, default=5_000_000)
    ap.add_argument("--keyframes", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    W, H, K, N = a.width, a.height, a.keyframes, a.splats
    res = {"device": torch.cuda.get_device_name(0), "splats": N, "keyframes": K, "image": [H, W]}
    intr = synth.intrinsic_for(W, H)

    # ---- loop closure: the 400 m "street" of config 5 --------------------------------------
    raw = synth.make_splats(N, W, H, n_views=K, seed=5, z_range=(2.0, 120.0))
    m = harness.SurfelMapper(raw, dev)
    c2ws = [p.to(dev) for p in synth.window_poses(K)]
    t = timed(lambda: maintenance.calc_score(m, c2ws, list(range(K)), intr), 1)
    res["calc_score_top3"] = entry(t, per_keyframe_ms=round(t / K, 3))
    scores3, ids3 = maintenance.calc_score(m, c2ws, list(range(K)), intr)
    new_c2ws = [c.clone() for c in c2ws]
    for k in range(K):
        new_c2ws[k][:3, 3] += 1.5 * k / K                       # drift spread along the loop
    scales = torch.ones(K)
    raw_stack, new_stack = torch.stack(c2ws), torch.stack(new_c2ws)
    t = timed(lambda: maintenance.rectify_gaussians(m, raw_stack, new_stack, scales, 0, K - 1, scores3,
                                                    ids3, delete_below=-1.0), a.reps)
    res["rectify_gaussians"] = entry(t, N * (3 + 4) * 4 * 2 + N * 4,
                                     note="re-anchor kernel + pose tables + the (empty) prune pass")
    dense = torch.zeros((K, H, W), device=dev)
    coarse = torch.zeros((K, H // 8, W // 8), device=dev)
    t = timed(lambda: maintenance.rectify_tracker(m, new_c2ws, intr, dense, coarse), 1)
    res["rectify_tracker"] = entry(t, per_keyframe_ms=round(t / K, 3))
    src, al = torch.rand(H, W, device=dev), torch.rand(H, W, device=dev)
    t = timed(lambda: _lib.check(L.vings_depth_refresh(H, W, _lib.ptr(src), _lib.ptr(al), _lib.ptr(dense[0]),
                                                       _lib.ptr(coarse[0]), _lib.stream_ptr(dev)), "r"), a.reps)
    res["depth_refresh_kernel"] = entry(t, H * W * 16 + H * W // 16)

    # ---- pruning compaction of the full state -------------------------------------------
    keep = torch.rand(N, device=dev) > 0.2
    arrays = [m.params[k].detach() for k in m.NAMES] + [m.global_scores, m.local_scores,
                                                        m.stable_mask, m.globalkf_id, m.globalkf_max_scores]
    row = sum(x[0].numel() * x.element_size() for x in arrays)
    t = timed(lambda: maintenance.compact(arrays, keep), a.reps)
    res["compact_params_and_records"] = entry(t, N * row + int(keep.sum()) * row + N)
    del m, raw

    # ---- keyframe arrival ---------------------------------------------------------------
    pts = torch.rand(40000, 3, device=dev) * torch.tensor([30.0, 5.0, 60.0], device=dev)
    t = timed(lambda: maintenance.mean_dist2_to_3nn(pts), a.reps)
    res["knn3_40000_points"] = entry(t, pair_tests=40000 * 40000)
    depth = torch.rand(1, H, W, device=dev) * 20 + 1
    rgb = torch.rand(3, H, W, device=dev)
    accum = torch.rand(1, H, W, device=dev)
    t = timed(lambda: maintenance.keyframe_pointcloud(c2ws[1], intr, rgb, depth, accum, 40000, 0.5), a.reps)
    res["keyframe_pointcloud"] = entry(t)

    # ---- tracker -> mapper packaging ---------------------------------------------------------
    ht, wd, Kf, B = H // 8, W // 8, 8, 12
    disps = torch.rand(B, ht, wd, device=dev) + 0.05
    cov = torch.rand(B, ht, wd, device=dev)
    up_d = torch.zeros(B, H, W, device=dev)
    up_c = torch.zeros(B, H, W, device=dev)
    mask = torch.randn(1, Kf, 576, ht, wd, device=dev)
    ix = torch.arange(2, 2 + Kf, device=dev)
    t = timed(lambda: TP.upsample_maps(ix, mask, [disps, cov], [up_d, up_c]), a.reps)
    res["upsample_disps_and_cov"] = entry(t, Kf * ht * wd * (576 * 4 + 2 * 4 + 2 * 64 * 4))
    P, t0 = 8, 2
    ii, jj = [], []
    for i in range(1, 2 + P + 1):
        for j in range(1, 2 + P + 1):
            if 1 <= abs(i - j) <= 3:
                ii.append(i); jj.append(j)
    ii, jj = np.array(ii), np.array(jj)
    M, Kd = len(ii), len(np.unique(ii))
    A = np.random.default_rng(0).normal(size=(6 * P, 6 * P))
    Hm = torch.from_numpy(A @ A.T + 6 * P * np.eye(6 * P))
    Q = torch.rand(Kd, ht * wd, device=dev) + 0.1
    E = torch.randn(P + M, 6, ht * wd, device=dev)
    dcov = torch.zeros(B, ht, wd, device=dev)
    t = timed(lambda: TP.compute_covariances(Hm, Q, E, ht, wd, ii, jj, t0, t0 + P, disps, dcov), a.reps)
    table, _ = TP.coupling_table(ii, jj, t0, t0 + P)
    blocks = int((table >= 0).sum())
    res["depth_covariance"] = entry(t, ht * wd * (blocks * 24 + Kd * 12),
                                    flops=2 * blocks * 6 * 6 * P * ht * wd, free_poses=P, depth_maps=Kd,
                                    note="includes the host factorisation of the 6P x 6P matrix")
    images = torch.randint(0, 256, (B, 3, H, W), device=dev, dtype=torch.uint8)
    cov_up = torch.rand(B, H, W, device=dev)
    ids = list(range(2, 2 + Kf))
    t = timed(lambda: TP.package_keyframes(images, up_d + 0.05, cov_up, ids, 40.0, 3.0), a.reps)
    res["package_keyframes"] = entry(t, Kf * H * W * (3 * 4 + 11 + 12 + 4 + 8))
    # ---- dense bundle adjustment, one ba(itrs=2) call end to end -------------------------------
    from vings_mono_amd import dba as dba_mod
    for name, (gh, gw) in (("ba_itrs2_43x77", (43, 77)), ("ba_itrs2_135x240", (135, 240))):
        prob = dba_mod.make_bench_problem(dev, gh, gw)
        t = timed(prob.run, a.reps)
        res[name] = entry(t, **prob.describe())
    # ---- the tracker's other two native calls, at the reference's own sizes -------------------
    E_, h1, w1 = 48, 43, 77
    coords = torch.stack(torch.meshgrid(torch.arange(w1, device=dev, dtype=torch.float32),
                                        torch.arange(h1, device=dev, dtype=torch.float32),
                                        indexing="xy"), 0)[None].repeat(E_, 1, 1, 1)
    coords = (coords + torch.randn_like(coords) * 3).contiguous()
    total = 0.0
    nbytes = 0
    for lvl in range(4):
        h2, w2 = h1 >> lvl, w1 >> lvl
        vol = torch.randn(E_, h1, w1, h2, w2, device=dev, dtype=torch.float16)
        c = (coords / 2 ** lvl).contiguous()
        total += timed(lambda: dba_mod.corr_index_forward(vol, c, 3), a.reps)
        nbytes += E_ * h1 * w1 * 49 * 2 * 5     # 49 outputs, four taps each, float16
        del vol
    res["corr_index_forward_4_levels"] = entry(total, nbytes, edges=E_, grid=[h1, w1], radius=3)
    prob = dba_mod.make_bench_problem(dev, h1, w1)
    ii = torch.arange(0, 8, device=dev).repeat_interleave(8)
    jj = torch.arange(0, 8, device=dev).repeat(8)
    t = timed(lambda: dba_mod.frame_distance(prob.poses, prob.disps, prob.intr, ii, jj, 0.3), a.reps)
    res["frame_distance_64_pairs"] = entry(t)
    # ---- on-the-fly correlation (no all-pairs volume) on the 1080p/8 grid ----------------------
    E_, h1, w1, C = 48, H // 8, W // 8, 128
    f1 = torch.randn(E_, h1, w1, C, device=dev)
    # two target fields, named apart: a tracker's flow is smooth (neighbouring pixels look at
    # neighbouring targets: the kernel's LDS-tiled path), targets scattered by 3 px per pixel
    # are the worst case (level 0 falls back to global reads)
    total = {"smooth_field_0.5px_jitter": 0.0, "scattered_field_3px_jitter": 0.0}
    total_b = 0.0
    for lvl in range(4):
        h2, w2 = h1 >> lvl, w1 >> lvl
        f2 = torch.randn(E_, h2, w2, C, device=dev)
        xy = torch.stack(torch.meshgrid(torch.arange(w1, device=dev, dtype=torch.float32),
                                        torch.arange(h1, device=dev, dtype=torch.float32),
                                        indexing="xy"), -1)[None, None].repeat(E_, 1, 1, 1, 1)
        for name, jitter in (("smooth_field_0.5px_jitter", 0.5), ("scattered_field_3px_jitter", 3.0)):
            c = ((xy + 2.5 + torch.randn_like(xy) * jitter) / 2 ** lvl).contiguous()
            total[name] += timed(lambda: dba_mod.altcorr_forward(f1, f2, c, 3), a.reps)
        g = torch.randn(E_, 1, 49, h1, w1, device=dev)
        total_b += timed(lambda: dba_mod.altcorr_backward(f1, f2, c, g, 3), a.reps)
        del f2
    for name, t in total.items():
        res["altcorr_forward_4_levels_" + name] = entry(t, edges=E_, grid=[h1, w1], channels=C, radius=3,
                                                        flops=2 * 4 * E_ * h1 * w1 * 64 * C)
    res["altcorr_backward_4_levels"] = entry(total_b, edges=E_, grid=[h1, w1], channels=C, radius=3)
    del f1

    # ---- the multi-GPU tail of a view step without the wire: collectives replaced by no-ops ----
    from vings_mono_amd import parallel

    class LocalReducer(parallel.GradReducer):
        def _all_reduce(self, t, op, async_op=False):
            return self._Done() if async_op else None

    n1 = 1_000_000
    raw = synth.make_splats(n1, W, H, n_views=8, seed=1)
    w2c = torch.linalg.inv(synth.window_poses(8)[3]).to(dev)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    with torch.no_grad():
        pr = harness.render(gtp, w2c, intr)
    gt = {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(), "depth_cov": torch.ones_like(pr["depth"])}
    cam, Rh = harness.view_settings(w2c, intr), w2c[:3, :3].cpu()
    tail = {}
    for name, red in (("single_gpu", None),
                      ("planar", LocalReducer(n1, dev, 2, compact=False)),
                      ("compact_4_chunks", LocalReducer(n1, dev, 2, n_chunks=4)),
                      ("compact_1_chunk", LocalReducer(n1, dev, 2, n_chunks=1))):
        mp_ = harness.SurfelMapper(raw, dev)
        for _ in range(2):
            mp_.fused_view(cam, intr, gt, Rh, 0, red)
        tail[name] = round(timed(lambda: mp_.fused_view(cam, intr, gt, Rh, 0, red), max(a.reps, 9)), 4)
        if red is not None and red.last_exchanged_rows is not None:
            tail[name + "_rows"] = red.last_exchanged_rows
        del mp_
    res["view_step_ms_with_exchange_tail_no_wire"] = dict(
        tail, splats=n1, note="whole fused view step; the difference to single_gpu is what the "
        "exchange adds per view besides the all-reduce time itself")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
