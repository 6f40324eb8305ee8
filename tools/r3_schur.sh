#!/bin/bash
# Parity of the DBA kernels, then per-kernel times of the DBA problem alone.
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout -k 10 400 python -u -m pytest tests/test_dba_parity.py tests/test_tracker_package_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3/schur_tests.log 2>&1 || { tail -30 gpurun_out/r3/schur_tests.log; exit 1; }
tail -3 gpurun_out/r3/schur_tests.log
for grid in 135x240 43x77; do
  out=gpurun_out/r3/dba_${grid}
  rm -rf $out
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/dba_only.py $grid 6 > $out.log 2>&1
  echo "== grid=$grid"
  python3 tools/kstats.py $out | grep -E "dba_" || true
done
