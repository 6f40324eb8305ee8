"""Times altcorr_forward (on-the-fly correlation window) over the four pyramid levels of a
1080p / 8 tracker grid, for a smooth flow field (what a tracker feeds) and for scattered targets.

    python tools/bench_altcorr.py [--edges 48] [--reps 5]   ->  one JSON line
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vings_mono_amd import dba as dba_mod  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def fields(E, h1, w1, dev, jitter, smooth):
    g = torch.Generator(device="cpu").manual_seed(11)
    xy = torch.stack(torch.meshgrid(torch.arange(w1, dtype=torch.float32),
                                    torch.arange(h1, dtype=torch.float32), indexing="xy"), -1)
    xy = xy[None, None].repeat(E, 1, 1, 1, 1)
    if smooth:
        # a translation per edge plus a slow shear: the induced flow of a camera motion
        t = (torch.rand(E, 1, 1, 1, 2, generator=g) - 0.5) * 30
        xy = xy + t + 0.02 * xy.flip(-1)
    xy = xy + torch.randn(xy.shape, generator=g) * jitter
    return xy.to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", type=int, default=48)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--height", type=int, default=135)
    ap.add_argument("--width", type=int, default=240)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    E, h1, w1, C = a.edges, a.height, a.width, 128
    f1 = torch.randn(E, h1, w1, C, device=dev)
    res = {}
    for name, jitter, smooth in (("smooth_jitter0.5", 0.5, True), ("scattered_jitter3", 3.0, False)):
        xy = fields(E, h1, w1, dev, jitter, smooth)
        per_level = []
        for lvl in range(4):
            f2 = torch.randn(E, h1 >> lvl, w1 >> lvl, C, device=dev)
            c = (xy / 2 ** lvl).contiguous()
            per_level.append(round(timed(lambda: dba_mod.altcorr_forward(f1, f2, c, 3), a.reps), 3))
        ms = sum(per_level)
        res[name] = {"ms": round(ms, 3), "per_level_ms": per_level,
                     "tflops": round(2 * 4 * E * h1 * w1 * 64 * C / ms / 1e9, 2)}
    print(json.dumps({"altcorr_forward_4_levels": res, "edges": E, "grid": [h1, w1], "channels": C,
                      "radius": 3}))


if __name__ == "__main__":
    main()
