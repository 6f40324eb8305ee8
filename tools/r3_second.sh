#!/bin/bash
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests/test_surfel_parity.py tests/test_splat_step_parity.py -x -q -m gpu --timeout 600 --timeout-method thread -k "stable or pixel_mask or without_scores or carry or finish or fused_view" > gpurun_out/r3/t2.log 2>&1 || { tail -60 gpurun_out/r3/t2.log; exit 1; }
tail -3 gpurun_out/r3/t2.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r3/bench_stable.json 2> gpurun_out/r3/bench_stable.err || { tail -30 gpurun_out/r3/bench_stable.err; exit 1; }
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r3/bench_stable.json').read().strip().splitlines()[-1])
print(r['value'], r['stage_ms'].get('blend_bwd'), r['stable_half'], r['full_passes']['value'], r['autograd_api']['value'])
PY
