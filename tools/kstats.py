#!/usr/bin/env python3
"""Print rows of a rocprofv3 kernel_stats.csv (first one found under a directory):
    python3 tools/kstats.py <dir-or-csv> [name-substring ...]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
pats = sys.argv[2:]
for r in csv.DictReader(open(path, newline="")):
    if not pats or any(p in r["Name"] for p in pats):
        print("%-56s calls %5s avg %9.1f us  min %8.1f  max %8.1f" % (
            r["Name"].split("(")[0][-56:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            float(r["MaxNs"]) / 1e3))
