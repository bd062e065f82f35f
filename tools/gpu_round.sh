cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06/gpu_tests.txt
cat gpurun_out/r06/gpu_tests.txt
python3 bench.py > gpurun_out/r06/default_bench_line.json 2> gpurun_out/r06/default.err; tail -c 300 gpurun_out/r06/default.err; cat gpurun_out/r06/default_bench_line.json | cut -c1-1500
bash tools/tick_ranks.sh after2 > /dev/null 2>&1; grep -E "^##|prepare us|^ranks" gpurun_out/tick_ranks_after2.txt
