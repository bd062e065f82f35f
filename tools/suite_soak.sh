#!/bin/bash
# The GPU test tier RUNS times in a row; the full output of every run that does not end with exit code 0 is kept
# (gpurun_out/suite_soak_<label>_<k>.log). Usage: tools/suite_soak.sh <label> [pytest args...]   (env: RUNS, default 10)
cd "$(dirname "$0")/.."
label=$1; shift
mkdir -p gpurun_out
bad=0
for k in $(seq 1 ${RUNS:-10}); do
    python -X faulthandler -m pytest tests -m gpu -x -q "$@" > /tmp/suite.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then bad=$((bad + 1)); cp /tmp/suite.log gpurun_out/suite_soak_${label}_$k.log; echo "run $k: exit code $rc"; else echo "run $k: ok ($(grep -E 'passed' /tmp/suite.log | tail -1))"; fi
done
echo "$label: $bad of ${RUNS:-10} runs of the GPU test tier failed"
