#!/bin/bash
# Evidence behind the measured numbers: kernel statistics of the bench command, HBM and SQ counter passes of the
# view step (default map and half of it stable), and counter passes of the dense-BA kernels.
# Every GPU step has its own time limit; the chain stops at the first failure.
set -o pipefail
OUT=gpurun_out/r3prof
mkdir -p $OUT
export TMPDIR=/tmp
SHORT="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dba --no-side-runs"
one() { ls $1/*/*counter_collection.csv | head -1; }

timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- \
    python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-side-runs > $OUT/bench_under_rocprof.json 2> $OUT/trace.err &&
cp "$(ls $OUT/trace/*/*kernel_stats.csv | head -1)" $OUT/bench_kernel_stats.csv &&
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $SHORT > /dev/null 2> $OUT/pmc_fetch.err &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $SHORT > /dev/null 2> $OUT/pmc_write.err &&
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $SHORT > /dev/null 2> $OUT/pmc_sq.err &&
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq_stable -- python3 $SHORT --stable-frac 0.5 > /dev/null 2> $OUT/pmc_sq_stable.err &&
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dba_trace -- python3 tools/dba_only.py 135x240 8 > /dev/null 2> $OUT/dba_trace.err &&
cp "$(ls $OUT/dba_trace/*/*kernel_stats.csv | head -1)" $OUT/dba_kernel_stats.csv &&
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/dba_fetch -- python3 tools/dba_only.py > /dev/null 2> $OUT/dba_fetch.err &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/dba_write -- python3 tools/dba_only.py > /dev/null 2> $OUT/dba_write.err &&
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/dba_sq -- python3 tools/dba_only.py > /dev/null 2> $OUT/dba_sq.err &&
{
  echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 $SHORT"
  echo "# mean counter value per launch, unit KiB; gfx950: FETCH_SIZE counts half of wide (16 B/lane) streaming reads"
  python3 tools/pmc_summary.py "$(one $OUT/pmc_fetch)"
  python3 tools/pmc_summary.py "$(one $OUT/pmc_write)"
} > $OUT/bench_pmc_hbm.txt &&
{
  echo "# rocprofv3 --pmc <8 SQ counters> -- python3 $SHORT ; mean per launch"
  python3 tools/pmc_summary.py "$(one $OUT/pmc_sq)" blend
  echo "# the same with --stable-frac 0.5 (half of the splats under the stable mask)"
  python3 tools/pmc_summary.py "$(one $OUT/pmc_sq_stable)" blend
} > $OUT/blend_sq_counters.txt &&
{
  echo "# rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ MFMA / VALU counters) -- python3 tools/dba_only.py"
  echo "# ba(itrs=2), 135x240 grid, 48 edges, 8 free poses; mean per launch; FETCH/WRITE in KiB"
  python3 tools/pmc_summary.py "$(one $OUT/dba_fetch)" dba_
  python3 tools/pmc_summary.py "$(one $OUT/dba_write)" dba_
  python3 tools/pmc_summary.py "$(one $OUT/dba_sq)" dba_
} > $OUT/dba_counters.txt &&
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_sq_stable $OUT/dba_trace $OUT/dba_fetch $OUT/dba_write $OUT/dba_sq &&
echo profiles-ok
