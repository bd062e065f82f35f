#!/bin/bash
# GPU box: the headless tick driver (CPU reference-path system vs the GPU drop-in, every buffer compared every tick) over many
# seeds and flag combinations — the mirror's maintenance paths under entity churn, re-parenting, toggles, moving scenes.
# Round 6: the multi-GPU mode kept current slot by slot (--unversioned: mesh systems without change counters; --animate-step: roots
# that cross cells take their trees to another rank). Round 5: the prepareMeshes gate (--gate: systems that are not ready / ready for some passes only / empty, each system also checked
# against the reference text; --skip-pass: a shadow pass left out by prepareShadowRender) and the drop-in's multi-GPU mode (--ranks N: one thread, N contexts; the rows travel by peer stores, with --communicator over the test transport).
#   tools/tick_soak.sh [SEEDS]      (default 40 seeds x 33 flag sets)
set -u
cd "$(dirname "$0")/.."
make -s -C tests/cpp
seeds=${1:-40}
bad=0; runs=0
sets=(
  "--entities 30000 --hier --mutate --churn 6"
  "--entities 20000 --mixed --hier --churn 5 --bounds"
  "--entities 40000 --mixed --csm --mutate"
  "--entities 12000 --animate 5 --hier --mixed --bounds --ticks 6"
  "--entities 9000 --animate 2 --span-records --churn 4 --ticks 5"
  "--entities 50000 --hier --toggle"
  "--entities 30000 --hier --animate 11 --itemised --world --mutate --ticks 4"
  "--entities 20000 --hier --mixed --animate 5 --itemised --world --churn 3 --ticks 3"
  "--entities 3000 --mutate --churn 9 --mixed"
  "--entities 70000 --churn 4 --copy-records"
  "--entities 33000 --mixed --csm --animate 7 --ticks 5 --soa-records"
  "--entities 300000 --animate 64 --churn 2 --ticks 3"
  "--entities 20000 --mixed --gate never --hier --mutate --churn 3"
  "--entities 20000 --mixed --gate shadow --animate 4 --ticks 4 --csm"
  "--entities 16000 --mixed --gate reverse --toggle --hier"
  "--entities 16000 --mixed --gate empty --churn 4 --span-records"
  "--entities 30000 --ranks 4 --hier --mutate --churn 4"
  "--entities 20000 --ranks 3 --mixed --hier --animate 5 --itemised --ticks 4"
  "--entities 24000 --ranks 8 --mixed --csm --churn 2"
  "--entities 12000 --ranks 2 --mixed --gate shadow --toggle --hier"
  "--entities 24000 --ranks 3 --mixed --unversioned --hier --mutate --churn 2"
  "--entities 20000 --ranks 4 --mixed --csm --unversioned --animate 3 --ticks 5"
  "--entities 16000 --ranks 4 --hier --animate 2 --animate-step 170 --itemised --ticks 7"
  "--entities 16000 --ranks 2 --hier --mixed --animate 3 --animate-step 230 --ticks 5 --churn 2"
  "--entities 20000 --ranks 3 --hiz --mixed --csm --unversioned --toggle --hier"
  "--entities 30000 --hier --mutate --churn 6 --same-frame"
  "--entities 20000 --mixed --hier --csm --churn 4 --same-frame --ranks 4"
  "--entities 40000 --hier --churn 8 --same-frame --ranks 3 --unversioned --ticks 2"
  "--entities 24000 --ranks 4 --mixed --csm --churn 2 --communicator"
  "--entities 20000 --ranks 3 --mixed --hier --animate 5 --itemised --ticks 4 --communicator"
  "--entities 16000 --ranks 2 --hier --mixed --animate 3 --animate-step 230 --ticks 5 --churn 2 --communicator"
  "--entities 20000 --mixed --csm --gate shadow --skip-pass 1 --churn 3"
  "--entities 16000 --mixed --gate reverse --skip-pass 0 --hier --mutate"
)
export GV_RCCL_LIBRARY=${GV_RCCL_LIBRARY:-$PWD/tests/cpp/build/librccl_stub.so}  # (--ranks N > 1: N contexts share this box's GPU)
for s in $(seq 1 "$seeds"); do
  for a in "${sets[@]}"; do
    runs=$((runs + 1))
    out=$(./tests/cpp/build/headless_tick --mode both --seed "$s" $a 2>&1 | tail -1)
    if ! echo "$out" | grep -q '"ok": true'; then bad=$((bad + 1)); echo "seed $s $a: $out" | cut -c1-600; fi
  done
done
echo "tick soak: $runs runs ($seeds seeds x ${#sets[@]} flag sets), $bad failed"
