#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel launch.

    python3 tools/pmc_summary.py gpurun_out/pmc/x_counter_collection.csv [name-filter]
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "vings::"
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if filt not in name:
                continue
            short = name.split("(")[0]
            acc[short][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[short].add(row["Dispatch_Id"])
    for k in sorted(acc):
        n = max(1, len(launches[k]))
        vals = "  ".join(f"{c}={v / n:.4g}" for c, v in sorted(acc[k].items()))
        print(f"{k} [{n} launches]  {vals}")


if __name__ == "__main__":
    main()
