cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=tests/cpp/build/headless_tick
for n in 2000 10000 100000; do $T --mode gpu --entities $n --ticks 500; done
$T --mode gpu --entities 10000 --ticks 500 --mixed
$T --mode cpu --entities 10000 --ticks 500 --avx2 --threads 1
$T --mode cpu --entities 10000 --ticks 500 --avx2 --threads 8
$T --mode cpu --entities 10000 --ticks 500 --avx2 --threads 32
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d gpurun_out/tick -- $T --mode gpu --entities 10000 --ticks 500 > gpurun_out/tick.log 2>&1
find gpurun_out/tick -name "*stats*" | head
