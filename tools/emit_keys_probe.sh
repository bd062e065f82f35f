#!/bin/bash
# GPU box (experiment, profiles/withdrawn.md 39): tools/sort_bench.py — cull + emit + gv_sort per frame at 10 M entities, hipEvents per
# kernel — with the emit as it is and with an emit that writes only (pool slot, distance key) per record and gathers only the position
# (GV_DEBUG_EMIT_KEYS_ONLY=1: the cheapest emit a "sort the pairs first, build the records afterwards" scheme could have; the records
# it leaves are not valid, only the times are).
for e in "" "GV_DEBUG_EMIT_KEYS_ONLY=1" "" "GV_DEBUG_EMIT_KEYS_ONLY=1"; do
  echo "## ${e:-emit as it is}"
  env $e timeout 300 python3 tools/sort_bench.py 2>&1 | grep records
done
