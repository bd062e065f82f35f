#!/bin/bash
# GPU box: same-box A/B of two builds of the library on the kernels that run the sphere pre-test beside other VALU work —
# cull_multi_kernel (main camera + 3 cascades, 10 M) and the fused sweep + cull kernels of cfg4 — plus the headline frames (cfg3, cfg2 at 10 M)
# as a no-regression check.   tools/classify_ab.sh OLD.so   (ROUNDS, default 3)
cd "$(dirname "$0")/.."
old=$(realpath "$1")
line() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d['roofline']
print('$1', '$2', 'ms/step %.4f' % d['ms_per_step'], 'median %.4f' % d['ms_per_step_median'], 'kernel %.1f us' % (r['avg_launch_ms'] * 1e3), 'frac %.3f' % r['frac'])"; }
for r in $(seq 1 ${ROUNDS:-3}); do
  for which in old new; do
    if [ $which = old ]; then export GV_LIB_PATH=$old; else unset GV_LIB_PATH; fi
    MULTIVIEW_QUICK=1 python tools/multiview_bench.py 2>/dev/null | grep batched | sed "s/^/$which multiview /"
    python bench.py --workload cfg4 --sweep fused --no-cpu-baseline --no-parity --steps 100 2>/dev/null | line $which cfg4-mfma
    python bench.py --workload cfg4 --sweep fused-valu --no-cpu-baseline --no-parity --steps 100 2>/dev/null | line $which cfg4-valu
    python bench.py --no-cpu-baseline --no-parity --no-hard-depth-variant 2>/dev/null | line $which cfg3
  done
done
