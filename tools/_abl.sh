EXTRA= bash tools/bench_summary.sh cfg3
EXTRA="--entities 10000000" bash tools/bench_summary.sh cfg2
EXTRA="--block-bounds" bash tools/bench_summary.sh cfg3
EXTRA= bash tools/bench_summary.sh cfg2
