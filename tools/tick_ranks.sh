#!/bin/bash
# GPU box: the drop-in's one-process multi-GPU mode (GpuVisibilitySystem with R contexts, all on this box's one GPU) — host
# microseconds per tick of the prepare phase, by step. The lists travel by peer stores (the drop-in's default for the devices of
# one process: real device copies, here within one GPU), or with `--communicator` through gv_exchange_init_all and the tests'
# host-staged transport (never a measurement of a link).
#   tools/tick_ranks.sh <label> [--communicator]   ->  gpurun_out/tick_ranks_<label>.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
label=${1:-run}
extra=${2:-}
out=gpurun_out/tick_ranks_$label.txt
mkdir -p gpurun_out
make -s -C tests/cpp >/dev/null 2>&1
export GV_RCCL_LIBRARY=$PWD/tests/cpp/build/librccl_stub.so
{
  echo "# tests/cpp/headless_tick --mode gpu --ranks 4 <args>, GV_TICK_BREAKDOWN=1 (host us per tick of the prepare phase, by step) — $label"
  echo "# (--churn R: R rounds, each destroys ~1 % and creates ~2 % of the entities, some under existing parents, then --ticks frames: the breakdown averages over all of them)"
  echo "# $(git rev-parse --short HEAD 2>/dev/null || echo snapshot) $(date -u +%FT%TZ)"
  for a in "--entities 10000 --mixed --csm --ticks 500" "--entities 10000 --mixed --csm --ticks 500 --unversioned" \
           "--entities 1000000 --mixed --csm --ticks 200" "--entities 1000000 --mixed --csm --ticks 100 --unversioned" \
           "--entities 10000 --mixed --csm --ticks 500 --unversioned --animate 50 --itemised" \
           "--entities 1000000 --mixed --csm --ticks 100 --unversioned --animate 50 --itemised" \
           "--entities 1000000 --ticks 200" "--entities 1000000 --ticks 100 --unversioned" \
           "--entities 100000 --mixed --hier --churn 30 --ticks 20" "--entities 1000000 --hier --churn 10 --ticks 10"; do
    echo "## --ranks 4 $a $extra"
    GV_TICK_BREAKDOWN=1 timeout 600 ./tests/cpp/build/headless_tick --mode gpu --ranks 4 $a $extra 2>&1 | grep -E "prepare us|exchanges|\"ok\"" | cut -c1-400
  done
  echo "## one context, for scale: --entities 10000 --mixed --csm --ticks 500 / --entities 1000000 --mixed --csm --ticks 200"
  GV_TICK_BREAKDOWN=1 timeout 600 ./tests/cpp/build/headless_tick --mode gpu --entities 10000 --mixed --csm --ticks 500 2>&1 | grep -E "prepare us"
  GV_TICK_BREAKDOWN=1 timeout 600 ./tests/cpp/build/headless_tick --mode gpu --entities 1000000 --mixed --csm --ticks 200 2>&1 | grep -E "prepare us"
} > $out 2>&1
cat $out
