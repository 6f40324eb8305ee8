#!/bin/bash
# multi-rank tests, the self-launching bench and the two side workloads in one GPU call
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests/test_multirank_gpu.py tests/test_maintenance_parity.py -x -q -m gpu --timeout 600 --timeout-method thread > gpurun_out/r3/t1.log 2>&1 || { tail -40 gpurun_out/r3/t1.log; exit 1; }
tail -3 gpurun_out/r3/t1.log
timeout -k 10 400 python bench.py > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err || { tail -30 gpurun_out/r3/bench_default.err; exit 1; }
timeout -k 10 300 python bench.py --workload native > gpurun_out/r3/bench_native.json 2> gpurun_out/r3/bench_native.err || { tail -30 gpurun_out/r3/bench_native.err; exit 1; }
timeout -k 10 300 python bench.py --scene surfaces --no-cpu-baseline > gpurun_out/r3/bench_surfaces.json 2> gpurun_out/r3/bench_surfaces.err || { tail -30 gpurun_out/r3/bench_surfaces.err; exit 1; }
echo done
