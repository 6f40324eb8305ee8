#!/bin/bash
# Sanity runs of the bench's other workloads on a one-GPU box: c4 alone, then c4 and c5 as
# two-rank gloo rehearsals at reduced size.  Not benchmark results.
set -o pipefail
OUT=gpurun_out/rehearse
mkdir -p $OUT
TR="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-side-runs"
timeout -k 10 300 python3 bench.py --workload c4 $COMMON > $OUT/c4_n1.json 2> $OUT/c4_n1.err &&
timeout -k 10 300 $TR --master-port 29801 bench.py --gpus 2 --rehearse-gloo --workload c4 --splats 200000 --width 960 --height 544 $COMMON > $OUT/c4_n2.json 2> $OUT/c4_n2.err &&
timeout -k 10 300 $TR --master-port 29802 bench.py --gpus 2 --rehearse-gloo --workload c5 --splats 500000 --loop-keyframes 8 --width 960 --height 544 $COMMON > $OUT/c5_n2.json 2> $OUT/c5_n2.err &&
timeout -k 10 300 $TR --master-port 29803 bench.py --gpus 2 --rehearse-gloo --workload c3 --exchange planar --splats 200000 --width 960 --height 544 $COMMON > $OUT/c3_planar_n2.json 2> $OUT/c3_planar_n2.err
r=$?
for f in c4_n1 c4_n2 c5_n2 c3_planar_n2; do echo "== $f"; cat $OUT/$f.json 2>/dev/null | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['n_gpus'], d['config']['workload'][:70], d['config'].get('exchange'))
except Exception as e: print('no json', e)
"; done
exit $r
