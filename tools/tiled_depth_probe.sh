#!/bin/bash
# GPU box: cfg3 on the hard depth image and on the walls, level-0 / virtual level-1 Hi-Z queries from the row-major depth image (default)
# against an extra copy of it in 8 x 8-texel tiles written by the first pyramid launch (GV_DEBUG_HIZ_TILED_DEPTH=1) — build + cull together.
for depth in noise walls; do
  for e in "" "GV_DEBUG_HIZ_TILED_DEPTH=1" "" "GV_DEBUG_HIZ_TILED_DEPTH=1"; do
    echo "## --depth $depth ${e:-default (row-major depth image)}"
    env $e python3 bench.py --depth $depth --no-cpu-baseline --no-hard-depth-variant --steps 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms_per_step %.4f  cull kernel %.4f ms  frac %.3f  visible %.4f  frame kernels %s  parity %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['visible_fraction'], d['config']['frame_kernel_ms'], d['parity']['visible_set_bit_identical']))"
  done
done
