#!/bin/bash
# the whole GPU suite in one process, slowest tests listed
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout -k 10 1100 python -u -m pytest tests -x -q -m gpu --timeout 600 --timeout-method thread --durations=15 > gpurun_out/r3/gputests.log 2>&1
rc=$?
tail -40 gpurun_out/r3/gputests.log
exit $rc
