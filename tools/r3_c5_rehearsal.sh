#!/bin/bash
# Config 5's loop closure dealt out over the ranks, rehearsed with 1, 2 and 4 ranks that share
# the one GPU of the box (gloo): the ranks' renders queue up on the same GPU, so the loop stages
# take about as long as with one rank when the keyframes are sharded, and `ranks` times as long
# when every rank renders all of them.  Not a benchmark.
set -o pipefail
OUT=gpurun_out/r3c5
mkdir -p $OUT
COMMON="--workload c5 --splats 1000000 --loop-keyframes 16 --width 960 --height 544 --steps 2 --warmup 1 --no-cpu-baseline --no-side-runs"
timeout -k 10 300 python3 bench.py $COMMON > $OUT/n1.json 2> $OUT/n1.err &&
timeout -k 10 300 python3 bench.py --gpus 2 --rehearse-gloo $COMMON > $OUT/n2.json 2> $OUT/n2.err &&
timeout -k 10 300 python3 bench.py --gpus 4 --rehearse-gloo $COMMON > $OUT/n4.json 2> $OUT/n4.err
r=$?
python3 - <<'PY'
import json
out = {"what": "bench.py --workload c5 --splats 1000000 --loop-keyframes 16 --width 960 --height 544, "
               "1 / 2 / 4 ranks sharing ONE GPU over gloo: per-rank stage times in ms",
       "ranks": {}}
for n in (1, 2, 4):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r3c5/n{n}.json") if l.startswith("{")][-1])
        out["ranks"][str(n)] = {k: d["stage_ms"].get(k) for k in ("loop_score_top3", "loop_reanchor", "loop_tracker_refresh")}
        out["ranks"][str(n)]["ranks_seen"] = d.get("ranks_seen")
    except Exception as e:
        out["ranks"][str(n)] = str(e)
json.dump(out, open("gpurun_out/r3c5/c5_sharded_loop_rehearsal.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
exit $r
