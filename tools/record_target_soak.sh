#!/bin/bash
# Soak of gv_pool_set_record_target (the engine's own combinedMeshes page-locked and written by the device): RUNS times each of
#   * the shim's tick with record targets on and the pools growing (headless_tick --churn: the vectors reallocate),
#   * the bare C-ABI lifetime cases of tools/record_target_probe.py,
#   * (FULL=1) the whole GPU test tier.
# Prints how many runs ended in anything but exit code 0 (an abort inside the runtime is what round 2 saw with page-locked
# APPLICATION memory on the mirror path: 3 of 10 suite runs).
cd "$(dirname "$0")/.."
RUNS=${RUNS:-10}
declare -A bad
run() {  # run <label> <command...>
    local label=$1; shift
    "$@" > /tmp/soak.out 2>&1
    local rc=$?
    if [ $rc -ne 0 ]; then bad[$label]=$(( ${bad[$label]:-0} + 1 )); echo "  $label: exit code $rc"; tail -3 /tmp/soak.out | sed 's/^/    /'; fi
}
for i in $(seq 1 $RUNS); do
    run "tick churn"        ./tests/cpp/build/headless_tick --mode both --entities 20000 --ticks 60 --churn 6
    run "tick churn mixed"  ./tests/cpp/build/headless_tick --mode both --entities 8000 --ticks 60 --churn 6 --mixed
    run "tick churn hier"   ./tests/cpp/build/headless_tick --mode both --entities 20000 --ticks 40 --churn 6 --hier --world
    run "c-abi grow"        python tools/record_target_probe.py grow 12
    run "c-abi replace"     python tools/record_target_probe.py replace 12
    run "c-abi early free"  python tools/record_target_probe.py early_free 12
    [ "${FULL:-0}" = 1 ] && run "gpu test tier" python -m pytest tests -m gpu -x -q
done
echo "record-target soak, $RUNS runs each:"
for label in "tick churn" "tick churn mixed" "tick churn hier" "c-abi grow" "c-abi replace" "c-abi early free" "gpu test tier"; do
    [ "$label" = "gpu test tier" ] && [ "${FULL:-0}" != 1 ] && continue
    echo "  $label: ${bad[$label]:-0} of $RUNS runs failed"
done
grep -h "lost registration" /tmp/soak.out 2>/dev/null | tail -1
