#!/bin/bash
# GPU box: the round's text evidence (sort probe, tick breakdown, default bench line) -> gpurun_out/evidence
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/evidence
rm -rf $out; mkdir -p $out
{
  echo "# tools/onesweep_probe (built with -DGV_SORT_TRACE): gv_sort, capacity 10 M slots; times in us"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DGV_SORT_TRACE -Igarden_amd/csrc tools/onesweep_probe.hip -o /tmp/onesweep_probe 2>/dev/null
  timeout 120 /tmp/onesweep_probe 2124723 10000000
  timeout 120 /tmp/onesweep_probe 308383 10000000
  echo
  echo "# tools/sort_bench.py (through the C-ABI, cull + emit + gv_sort per frame, 50 frames; hipEvents)"
  timeout 200 python3 tools/sort_bench.py 2>&1 | grep records
  echo
  echo "# ranking variants measured on the way (same probe, same sizes; kept: the first)"
  echo "#   8-ballot match + wave-private LDS running counts (kept)            loads+rank 6.4-7.0 us/tile median, 112-118 VGPRs, gv_sort 223-231 us"
  echo "#   LDS lane-mask table (atomicOr + read back) instead of the ballots  loads+rank 6.4-9.3 us, no gain: the dependent LDS round trips replace the ALU"
  echo "#   4 rotating mask tables + returning LDS atomics for all 16 rounds   loads+rank 5.4-8.7 us but 9-12 VGPR spills at 128 VGPRs, reorder 0.9 -> 2.4-3.5 us, gv_sort 238-253 us"
  echo "#   static tile ids (blockIdx.x) instead of tickets                    gv_sort 223 -> 211 us; correct only under in-order dispatch: not kept"
} > $out/r02_sort_probe.txt 2>&1
{
  echo "# tests/cpp/headless_tick --mode gpu --ticks 2000 <args>, GV_TICK_BREAKDOWN=1 (host us per tick of the drop-in's prepare phase)"
  for a in "--entities 2000" "--entities 10000" "--entities 100000" "--entities 10000 --mixed" "--entities 10000 --mixed --csm" "--entities 10000 --hier --world --animate 50 --itemised"; do
    echo "## $a"
    GV_TICK_BREAKDOWN=1 ./tests/cpp/build/headless_tick --mode gpu --ticks 2000 $a 2>&1 | grep -E "prepare us"
  done
  echo "# round 1 (profiles/r01k_tick_*, DESIGN.md): 2 k 32-35, 10 k 49-52, 100 k 263, 10 k --mixed 210, --mixed --csm 203"
} > $out/r02_tick.txt 2>&1
python3 bench.py > $out/r02k_default_bench_line.json 2> $out/default.err
tail -c 600 $out/default.err
