#!/bin/bash
# GPU box: cfg2 (flat, frustum-only) at several sizes: cull + emit as two launches (default), in one launch with the look-back chain
# (GV_DEBUG_FUSED_EMIT_MAX) and in one launch with the two-level count sums (+ GV_DEBUG_FUSED_TWO_LEVEL_MIN=0). Parity is bench.py's own.
for n in 20000 100000 300000 1000000 2000000 10000000; do
  for e in "GV_DEBUG_FUSED_EMIT_MAX=0" "GV_DEBUG_FUSED_EMIT_MAX=16777216" "GV_DEBUG_FUSED_EMIT_MAX=16777216 GV_DEBUG_FUSED_TWO_LEVEL_MIN=0"; do
    echo "## entities $n  $e"
    env $e python3 bench.py --workload cfg2 --entities $n --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms_per_step %.4f median %.4f  cull kernel %.4f ms  frame kernels %s parity %s' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'], d['config']['frame_kernel_ms'], d['parity']['visible_set_bit_identical'] and d['parity']['baked_model_bit_identical']))"
  done
done
