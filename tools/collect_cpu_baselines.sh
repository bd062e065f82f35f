#!/bin/bash
# GPU box: the CPU baseline beside EVERY BASELINE.json config, in kept files -> gpurun_out/baselines
#   cfg1: the reference CPU path itself — headless ecsm tick, 10 k entities, scalar / AVX2, 1 thread / all the box's CPUs
#   cfg2, cfg3, cfg4: bench.py --workload <cfg> with its cpu_baseline leg on (cfg4: threaded scalar chain sweep + AVX2 cull; sweep_ms > 0)
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/baselines
rm -rf $out; mkdir -p $out
for w in cfg2 cfg3 cfg4; do
  python3 bench.py --workload $w > $out/r05_${w}_bench_line.json 2> $out/${w}.err
done
threads=$(python3 -c "import bench; print(bench.effective_cores()[0])")
{
  echo "# cfg1 (BASELINE.json configs[0]): tests/cpp/headless_tick --mode cpu --entities 10000 --ticks 2000 — the CPU reference-path system"
  echo "# (oracle/cpu_mesh_render_system.hpp: prepareMeshes, mesh.cpp:331-553, over the scalar / AVX2 oracle), culls per second of wall clock;"
  echo "# beside it the GPU drop-in on the same tick (--mode gpu). all = $threads threads (the CPUs this process may use)"
  for a in "" "--avx2" "--threads $threads" "--avx2 --threads $threads"; do
    echo "## cpu ${a:-scalar, 1 thread}"
    ./tests/cpp/build/headless_tick --mode cpu --entities 10000 --ticks 2000 $a
  done
  echo "## gpu drop-in"
  ./tests/cpp/build/headless_tick --mode gpu --entities 10000 --ticks 2000
  echo "## gpu drop-in, --span-records"
  ./tests/cpp/build/headless_tick --mode gpu --entities 10000 --ticks 2000 --span-records
} > $out/r05_cfg1_tick.txt 2>&1
python3 - <<'PY' > gpurun_out/baselines/r05_cpu_baselines.txt
import json
print("# the CPU baseline beside every config (same run, same box): culls/s of the whole frame; cores = CPUs the process may use")
print("config  gpu_culls/s  gpu_ms/frame  cpu_culls/s  cpu_frame_ms  (cull / pyramid / sweep ms)  cores  threads(cull/pyramid/sweep)  cpu_avx2_1t_culls/s  gpu/cpu")
for w in ("cfg2", "cfg3", "cfg4"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/baselines/r05_{w}_bench_line.json") if l.startswith("{")][-1])
    except Exception as e:
        print(w, "missing:", e)
        continue
    c = d["cpu_baseline"]
    t = c["threads_used"]
    print(f"{w}  {d['value']:.3e}  {d['ms_per_step']:.4f}  {c['value']:.3e}  {c['frame_ms']:.2f}  ({c['cull_ms']:.2f} / {c['pyramid_ms']:.2f} / {c['sweep_ms']:.2f})  "
          f"{c['cores']}  {t['cull']}/{t['pyramid']}/{t['sweep']}  {c['avx2_soa_cull_1_thread_culls_per_s']:.3e}  {d['value'] / c['value']:.0f}")
PY
cat gpurun_out/baselines/r05_cpu_baselines.txt gpurun_out/baselines/r05_cfg1_tick.txt
