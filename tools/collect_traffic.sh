#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes for the cull kernel, one counter group per pass, as
# /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass).
# cfg2 @ 10M (frustum-only, known 72 B/entity stream) is the calibration run for this access pattern.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc
mkdir -p gpurun_out/pmc
for wl in "cfg3" "cfg2 --entities 10000000" "cfg3 --block-bounds" "cfg4" "cfg3 --depth noise"; do
  tag=$(echo $wl | cut -d" " -f1); case "$wl" in *block-bounds*) tag=${tag}bb;; *noise*) tag=${tag}hard;; esac
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc/${tag}_$c -- python3 bench.py --workload $wl --no-cpu-baseline --no-parity --steps 5 --warmup 2 > gpurun_out/pmc/${tag}_$c.log 2>&1
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/stats_cfg3 -- python3 bench.py --no-cpu-baseline --no-parity > gpurun_out/pmc/stats_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/stats_cfg3hard -- python3 bench.py --depth noise --no-cpu-baseline --no-parity > gpurun_out/pmc/stats_cfg3hard.log 2>&1
# the driver's own command under the profiler
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/stats_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/pmc/stats_driver.log 2>&1
env | grep -i -E "rocprof|LD_PRELOAD" > gpurun_out/pmc/plain_env.txt; rocprofv3 --kernel-trace -d gpurun_out/pmc/envprobe -- python3 -c "import os; print({k: v for k, v in os.environ.items() if 'ROCP' in k.upper() or k == 'LD_PRELOAD'})" > gpurun_out/pmc/profiler_env.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/stats_cfg4 -- python3 bench.py --workload cfg4 --no-cpu-baseline --no-parity > gpurun_out/pmc/stats_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/stats_cfg2 -- python3 bench.py --workload cfg2 --no-cpu-baseline --no-parity > gpurun_out/pmc/stats_cfg2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/stats_cfg3bb -- python3 bench.py --block-bounds --no-cpu-baseline --no-parity > gpurun_out/pmc/stats_cfg3bb.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/stats_cfg4valu -- python3 bench.py --workload cfg4 --sweep fused-valu --no-cpu-baseline --no-parity > gpurun_out/pmc/stats_cfg4valu.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/stats_cfg2_10M -- python3 bench.py --workload cfg2 --entities 10000000 --no-cpu-baseline --no-parity > gpurun_out/pmc/stats_cfg2_10M.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/stats_cfg5shape -- python3 bench.py --workload cfg5 --no-cpu-baseline --no-parity > gpurun_out/pmc/stats_cfg5shape.log 2>&1
# the driver's own command (N = 1 default, with the CPU baseline and the parity gate), not under the profiler
python3 bench.py > gpurun_out/pmc/stats_default.log 2> gpurun_out/pmc/default.err
ls gpurun_out/pmc
