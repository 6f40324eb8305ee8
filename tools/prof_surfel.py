#!/usr/bin/env python3
"""Profiling driver: rasteriser forward + backward only (no loss, no optimiser) at a
BASELINE-sized scene, for rocprofv3 kernel traces and PMC passes.

    rocprofv3 --kernel-trace --stats -- python3 tools/prof_surfel.py --iters 3
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splats", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--stats", action="store_true", help="print tile-list statistics")
    args = ap.parse_args()
    from vings_mono_amd import surfel, synth
    dev = torch.device("cuda:0")
    W, H = args.width, args.height
    raw = synth.make_splats(args.splats, W, H, n_views=args.views)
    act = {k: v.to(dev).contiguous() for k, v in synth.activate(raw).items()}
    c2w = synth.window_poses(args.views)[args.views // 2]
    cam = synth.Camera(torch.linalg.inv(c2w).to(dev), synth.intrinsic_for(W, H))
    settings = surfel.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=0, campos=cam.camera_center,
        prefiltered=False, debug=False)
    gC = torch.randn(3, H, W, device=dev) / (H * W)
    gA = torch.randn(9, H, W, device=dev) / (H * W)
    torch.cuda.synchronize()
    for it in range(args.iters):
        t0 = time.perf_counter()
        st, aux = surfel.rasterize_forward(act["means3D"], act["colors"], act["opacities"],
                                           act["scales"], act["rotations"], settings)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        surfel.rasterize_backward(st, aux, act["means3D"], act["scales"], act["rotations"], 1.0,
                                  gC, gA)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"iter {it}: forward {1e3 * (t1 - t0):.3f} ms  backward {1e3 * (t2 - t1):.3f} ms  "
              f"D={st.D}", flush=True)
    if args.stats:
        L = (st.ranges[:, 1] - st.ranges[:, 0]).float()
        nc = st.n_contrib[0].float()
        print("tile list: mean %.1f max %.0f ; last contributor: mean %.1f max %.0f ; visible %d"
              % (L.mean().item(), L.max().item(), nc.mean().item(), nc.max().item(),
                 int((st.radii > 0).sum().item())))
        g = st.geom[st.radii > 0]
        bw, bh = (g[:, 21] - g[:, 20]), (g[:, 23] - g[:, 22])
        print("alpha box: median %.1f x %.1f px, mean area %.0f px^2; 3-sigma radius median %.0f"
              % (bw.median().item(), bh.median().item(), (bw * bh).clamp(max=1e6).mean().item(),
                 st.radii[st.radii > 0].float().median().item()))


if __name__ == "__main__":
    main()
