#!/bin/bash
# prints a compact summary line per workload (used during tuning; not part of the product)
for w in "$@"; do python bench.py --workload $w --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:5], 'ms/step', round(d['ms_per_step'],4), 'value', '%.3g'%d['value'], {k: round(v*1000,1) for k,v in d['config']['kernel_ms'].items()}, 'frac', round(d['roofline']['frac'],3), 'parity', d['parity']['visible_set_bit_identical'] and d['parity']['baked_model_bit_identical'])"; done
