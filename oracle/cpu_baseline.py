"""CPU baseline leg of bench.py: the oracle timed on the host cores.

TEST INFRASTRUCTURE ONLY (see surfel_oracle.c).  The reference has no CPU path
(hard-coded device="cuda", /root/reference/scripts/gaussian/gaussian_base.py:145,158),
so the reported baseline is this port: kind = "port".

Sample: views of the benchmark window at full size (projection, binning, blend forward,
blend + projection backward, sparse Adam; OpenMP over the host cores), taken one after the
other until `budget_s` seconds of CPU work have been spent (at least two views, at most the
window); the dense-BA oracle (numpy float64) is timed once and amortised over the window's
views like in the GPU metric.  The loss between forward and backward is not part of the
sample (the oracle for it is a PyTorch graph, not a port), which favours the CPU number.
"""
import os
import time

import numpy as np
import torch


def run(n_splats, W, H, n_views, dba_problem=None, dba_itrs=2, max_threads=16, budget_s=12.0,
        window=8):
    from oracle.surfel_oracle import SurfelOracle, build
    from vings_mono_amd import synth
    threads = max(1, min(max_threads, os.cpu_count() or 1))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    build()
    orc = SurfelOracle(np.float32)
    raw = synth.make_splats(n_splats, W, H, n_views=n_views, seed=synth.SEED_BASE)
    act = {k: v.numpy() for k, v in synth.activate(raw).items()}
    poses = synth.window_poses(n_views)
    gC = np.full((3, H, W), 1.0 / (3 * H * W), np.float32)
    gA = np.full((9, H, W), 1.0 / (H * W), np.float32)
    groups = (("means3D", "means3D", 1.6e-4), ("colors", "colors", 2.5e-3),
              ("opacities", "opacity", 5e-2), ("scales", "scales", 5e-3),
              ("rotations", "rots", 1e-3))
    params = {n: np.ascontiguousarray(act[n], np.float32).reshape(n_splats, -1).copy()
              for n, _, _ in groups}
    m1 = {n: np.zeros_like(p) for n, p in params.items()}
    m2 = {n: np.zeros_like(p) for n, p in params.items()}
    t_f = t_b = t_a = 0.0
    done = 0
    order = [(n_views // 2 + i) % n_views for i in range(min(n_views, window))]
    for k in order:
        cam = synth.Camera(torch.linalg.inv(poses[k]), synth.intrinsic_for(W, H))
        view = cam.world_view_transform.numpy().reshape(16)
        proj = cam.full_proj_transform.numpy().reshape(16)
        t0 = time.perf_counter()
        st = orc.forward(act["means3D"], act["scales"], act["rotations"], act["opacities"],
                         act["colors"], view, proj, np.zeros(3, np.float32), H, W)
        t1 = time.perf_counter()
        g = orc.backward(st, gC, gA)
        t2 = time.perf_counter()
        vis = st["radii"] > 0
        for name, key, lr in groups:
            gr = np.ascontiguousarray(g[key], np.float32).reshape(n_splats, -1)
            orc.sparse_adam(params[name], gr, m1[name], m2[name], vis, lr)
        t3 = time.perf_counter()
        t_f += t1 - t0; t_b += t2 - t1; t_a += t3 - t2
        done += 1
        if done >= 2 and t_f + t_b + t_a >= budget_s:
            break
    dba_s = 0.0
    if dba_problem is not None:
        from oracle import dba_oracle
        keys = ("poses", "disps", "intr", "disps_sens", "target", "weight", "eta", "ii", "jj",
                "t0", "t1")
        td = time.perf_counter()
        dba_oracle.ba(*[dba_problem[k] for k in keys], itrs=dba_itrs, lm=1e-4, ep=0.1)
        dba_s = time.perf_counter() - td
    per_view = (t_f + t_b + t_a) / done + dba_s / window
    return {"value": round(1.0 / per_view, 5), "unit": "frames/s", "cores": threads,
            "kind": "port",
            "sample": f"{done} of {n_views} views at full size ({n_splats} splats, {W}x{H}), "
                      f"per view: forward {t_f / done:.2f}s, backward {t_b / done:.2f}s, "
                      f"adam {t_a / done:.2f}s (C oracle, OpenMP)"
                      + (f"; dense BA {dba_s:.2f}s per window of {window} views (numpy f64, "
                         f"1 core + BLAS threads)" if dba_problem is not None else
                         "; dense BA not included"),
            "seconds": round((t_f + t_b + t_a) + dba_s, 2)}
