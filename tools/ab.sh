#!/bin/bash
# Same-box A/B of one debug switch on the bench frame: tools/ab.sh ENV_NAME [bench args...] — runs bench.py alternately without and with
# ENV_NAME=1 (ROUNDS times each) and prints ms per step, the cull kernel and the per-kernel breakdown of a frame.
cd "$(dirname "$0")/.."
var=$1; shift
for r in $(seq 1 ${ROUNDS:-2}); do
  for on in 0 1; do
    if [ $on = 1 ]; then export $var=1; else unset $var; fi
    python bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config']
print('$var=$on', 'ms/step %.4f' % d['ms_per_step'], 'median %.4f' % d['ms_per_step_median'], 'value %.3e' % d['value'],
      {k: round(v * 1000, 1) for k, v in (c.get('frame_kernel_ms') or {}).items()}, 'bb %.4f cull %.1f us examined %.3f' % tuple((c.get('block_bounds_variant') or {}).get(k, 0) * f for k, f in (('ms_per_step', 1), ('cull_kernel_ms', 1e3), ('examined_workgroup_fraction', 1))))"
  done
done
