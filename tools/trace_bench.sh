#!/bin/bash
# rocprofv3 kernel statistics of a short default bench run -> gpurun_out/<name>/ks.csv, and the
# rows of the kernels named by the remaining arguments (substring match) on stdout.
# Usage: bash tools/trace_bench.sh NAME [kernel-substring ...]
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-runs --no-dba > $OUT/bench.json 2> $OUT/bench.err || exit 1
cp "$(ls $OUT/trace/*/*kernel_stats.csv | head -1)" $OUT/ks.csv && rm -rf $OUT/trace
python3 - "$OUT/ks.csv" "$@" <<'PY'
import csv, sys
pats = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    if not pats or any(p in r["Name"] for p in pats):
        print("%-64s calls %4s avg %9.1f us  min %8.1f  max %8.1f" % (
            r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
python3 -c "
import json,sys;d=json.load(open('$OUT/bench.json'));print(d['value'],d['stage_ms'])"
