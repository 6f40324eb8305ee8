#!/usr/bin/env python3
"""Idle time between consecutive GPU kernels from a rocprofv3 --kernel-trace CSV: prints, for the
steady-state part of the trace, the mean gap that follows each kernel name and the totals.

    python3 tools/gap_report.py <..._kernel_trace.csv> [skip_fraction]
"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1], newline="")))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0])
                 for r in rows), key=lambda e: e[0])
    ev = ev[int(len(ev) * skip):]
    gaps = defaultdict(lambda: [0, 0.0])
    busy = idle = 0.0
    one_off = []
    for a, b in zip(ev, ev[1:]):
        busy += a[1] - a[0]
        g = max(0, b[0] - a[1])
        if g > 100e3:
            # not a launch gap: the host was doing something else (set-up, a read-back, the end
            # of the timed region); listed, not counted
            one_off.append((g, a[2] + " -> " + b[2]))
            continue
        idle += g
        gaps[a[2] + " -> " + b[2]][0] += 1
        gaps[a[2] + " -> " + b[2]][1] += g
    print(f"kernels {len(ev)}  busy {busy / 1e6:.3f} ms  idle {idle / 1e6:.3f} ms  ({100 * idle / (busy + idle):.1f} % idle)")
    for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{t / 1e3:9.1f} us total  {t / n / 1e3:7.2f} us mean  x{n:<5d} {k}")
    for g, k in sorted(one_off, reverse=True)[:5]:
        print(f"(not counted: {g / 1e3:9.1f} us  {k[:120]})")


if __name__ == "__main__":
    main()
