#!/bin/bash
# A/B of two builds on the config-5 workload: the loop-closure stages.
set -o pipefail
for rep in 1 2; do
  for v in A B; do
    VINGS_HIP_LIB=$PWD/tools/experiments/ab/lib$v.so timeout -k 10 200 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['stage_ms']['loop_score_top3'], d['stage_ms']['loop_tracker_refresh'])" || exit 1
  done
done
