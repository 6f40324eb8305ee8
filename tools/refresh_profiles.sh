#!/bin/bash
# Collects, on the GPU box, everything the profiles/ directory holds for the default bench
# command: the bench line, rocprofv3 kernel statistics of the same command, the HBM counter
# passes (FETCH_SIZE and WRITE_SIZE in runs of their own) and the SQ passes of the blend
# kernels.  Writes under gpurun_out/refresh/; copy what is to be kept into profiles/.
# Every GPU step has its own time limit and the chain stops at the first failure.
set -o pipefail
OUT=gpurun_out/refresh
mkdir -p $OUT
export TMPDIR=/tmp
SHORT="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dba --no-side-runs"

timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- \
    python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-side-runs > $OUT/bench_under_rocprof.json 2> $OUT/trace.err &&
cp "$(ls $OUT/trace/*/*kernel_stats.csv | head -1)" $OUT/bench_kernel_stats.csv &&
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $SHORT > /dev/null 2> $OUT/pmc_fetch.err &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $SHORT > /dev/null 2> $OUT/pmc_write.err &&
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $SHORT > /dev/null 2> $OUT/pmc_sq.err &&
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_sq2 -- python3 $SHORT > /dev/null 2> $OUT/pmc_sq2.err &&
{
  echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 $SHORT"
  echo "# mean counter value per launch, unit KiB; gfx950: FETCH_SIZE counts half of wide (16 B/lane) streaming reads"
  python3 tools/pmc_summary.py "$(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1)"
  python3 tools/pmc_summary.py "$(ls $OUT/pmc_write/*/*counter_collection.csv | head -1)"
} > $OUT/bench_pmc_hbm.txt &&
{
  echo "# rocprofv3 --pmc <8 SQ counters> (two passes) -- python3 $SHORT ; mean per launch"
  python3 tools/pmc_summary.py "$(ls $OUT/pmc_sq/*/*counter_collection.csv | head -1)" blend
  python3 tools/pmc_summary.py "$(ls $OUT/pmc_sq2/*/*counter_collection.csv | head -1)" blend
} > $OUT/blend_sq_counters.txt &&
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_sq2 &&
echo refresh-ok
