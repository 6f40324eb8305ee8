#!/bin/bash
# A/B of two builds of the library on one box: alternating bench runs, blend stage times.
set -o pipefail
for rep in 1 2 3; do
  for v in A B; do
    VINGS_HIP_LIB=$PWD/tools/experiments/ab/lib$v.so timeout -k 10 150 python bench.py --no-cpu-baseline --no-side-runs 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['stage_ms']['blend_fwd'], d['stage_ms']['blend_bwd'], d['stage_ms']['bin'], d['stage_ms']['project_fwd'], d['stage_ms']['loss'])" || exit 1
  done
done
