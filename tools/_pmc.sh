cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sq; mkdir -p gpurun_out/sq
for wl in "cfg3" "cfg2 --entities 10000000"; do tag=$(echo $wl | cut -d" " -f1)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d gpurun_out/sq/$tag -- python3 bench.py --workload $wl --no-cpu-baseline --no-parity --steps 5 --warmup 2 > gpurun_out/sq/$tag.log 2>&1
done
ls gpurun_out/sq/*/*/ | head
