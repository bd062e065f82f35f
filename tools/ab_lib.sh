#!/bin/bash
# Same-box A/B of two builds of the library on the bench frame: tools/ab_lib.sh OLD.so [bench args...] — runs bench.py alternately with
# GV_LIB_PATH=OLD.so and with the in-tree library (ROUNDS times each), prints ms per step and the per-kernel breakdown of a frame.
cd "$(dirname "$0")/.."
old=$1; shift
for r in $(seq 1 ${ROUNDS:-3}); do
  for which in old new; do
    if [ $which = old ]; then export GV_LIB_PATH=$old; else unset GV_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config']
print('$which', 'ms/step %.4f' % d['ms_per_step'], 'median %.4f' % d['ms_per_step_median'], 'value %.3e' % d['value'],
      {k: round(v * 1000, 1) for k, v in (c.get('frame_kernel_ms') or {}).items()}, 'bb %.4f' % ((c.get('block_bounds_variant') or {}).get('ms_per_step', 0)))"
  done
done
