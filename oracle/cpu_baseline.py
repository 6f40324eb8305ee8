"""CPU baseline leg of bench.py: the oracle timed on the host cores.

TEST INFRASTRUCTURE ONLY (see surfel_oracle.c).  The reference has no CPU path
(hard-coded device="cuda", /root/reference/scripts/gaussian/gaussian_base.py:145,158),
so the reported baseline is this port: kind = "port".

Sample: ONE view of the benchmark window at full size (projection, binning,
blend forward, blend + projection backward, sparse Adam), OpenMP over the
host cores; the dense-BA oracle (numpy) is timed once and amortised over the
window like in the GPU metric.
"""
import os
import time

import numpy as np
import torch


def run(n_splats, W, H, n_views, dba_problem=None, max_threads=16):
    from oracle.surfel_oracle import SurfelOracle, build
    from vings_mono_amd import synth
    threads = max(1, min(max_threads, os.cpu_count() or 1))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    build()
    orc = SurfelOracle(np.float32)
    raw = synth.make_splats(n_splats, W, H, n_views=n_views, seed=synth.SEED_BASE)
    act = {k: v.numpy() for k, v in synth.activate(raw).items()}
    k = n_views // 2
    w2c = torch.linalg.inv(synth.window_poses(n_views)[k])
    cam = synth.Camera(w2c, synth.intrinsic_for(W, H))
    view = cam.world_view_transform.numpy().reshape(16)
    proj = cam.full_proj_transform.numpy().reshape(16)
    gC = np.full((3, H, W), 1.0 / (3 * H * W), np.float32)
    gA = np.full((9, H, W), 1.0 / (H * W), np.float32)
    t0 = time.perf_counter()
    st = orc.forward(act["means3D"], act["scales"], act["rotations"], act["opacities"],
                     act["colors"], view, proj, np.zeros(3, np.float32), H, W)
    t1 = time.perf_counter()
    g = orc.backward(st, gC, gA)
    t2 = time.perf_counter()
    vis = st["radii"] > 0
    for name, key, lr in (("means3D", "means3D", 1.6e-4), ("colors", "colors", 2.5e-3),
                          ("opacities", "opacity", 5e-2), ("scales", "scales", 5e-3),
                          ("rotations", "rots", 1e-3)):
        p = np.ascontiguousarray(act[name], np.float32).reshape(n_splats, -1).copy()
        gr = np.ascontiguousarray(g[key], np.float32).reshape(n_splats, -1)
        orc.sparse_adam(p, gr, np.zeros_like(p), np.zeros_like(p), vis, lr)
    t3 = time.perf_counter()
    dba_s = 0.0
    if dba_problem is not None:
        from oracle import dba_oracle
        td = time.perf_counter()
        dba_oracle.ba(**dba_problem)
        dba_s = time.perf_counter() - td
    per_view = (t3 - t0) + dba_s / 8.0
    return {"value": round(1.0 / per_view, 5), "unit": "frames/s", "cores": threads,
            "kind": "port",
            "sample": f"1 of {n_views} views at full size ({n_splats} splats, {W}x{H}): "
                      f"forward {t1 - t0:.2f}s, backward {t2 - t1:.2f}s, adam {t3 - t2:.2f}s"
                      + (f", dense BA {dba_s:.2f}s per window / 8 views" if dba_problem else
                         ", dense BA not included"),
            "seconds": round(per_view, 3)}
