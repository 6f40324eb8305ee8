#!/bin/bash
# Launch gaps at the reference's own operating point (344x616, 300k splats, 43x77 DBA grid):
# kernel trace of the eager view steps and of the same steps replayed from HIP graphs.
set -o pipefail
OUT=gpurun_out/r3gaps
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 python3 bench.py --workload native > $OUT/bench_native.json 2> $OUT/bench_native.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/eager -- \
    python3 bench.py --workload native --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs --no-dba > /dev/null 2> $OUT/eager.err || exit 1
{
  echo "# eager fused view steps, --workload native --no-dba (rocprofv3 --kernel-trace), second half of the trace"
  python3 tools/gap_report.py "$(ls $OUT/eager/*/*kernel_trace.csv | head -1)" 0.3
} > $OUT/native_gap_report.txt
rm -rf $OUT/eager
cat $OUT/native_gap_report.txt | head -12
