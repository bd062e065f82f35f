#!/bin/bash
# GPU box: bench.py --workload cfg2 (flat, frustum-only) at 10^4 .. 10^8 entities on ONE GPU, each size WITH the CPU baseline of the
# same run (AVX2+FMA SoA cull on the box's host cores, one thread and all the CPUs the process may use) -> gpurun_out/sweep
# (BASELINE north_star: "throughput on synthetic scenes of 10^4-10^8 entities ... as absolute numbers and as fraction of the HBM-read
# roofline, next to the reference AVX2 path timed on the same box's host cores in the same run (core count stated)").
# 10^8: the CPU side holds the AoS pools (12.8 GB), the SoA copy (7.3 GB) and the oracle's outputs — run with the baseline only when
# the box has the memory (MemAvailable >= 96 GB), without it otherwise; the table says which.
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sweep
rm -rf $out; mkdir -p $out
avail_gb=$(awk '/MemAvailable/ {printf "%d", $2 / 1048576}' /proc/meminfo)
echo "MemAvailable ${avail_gb} GB" > $out/box.txt
for n in 10000 100000 1000000 10000000 100000000; do
  steps=200; [ $n -ge 100000000 ] && steps=20
  flags=""
  if [ $n -ge 100000000 ] && [ "$avail_gb" -lt 96 ]; then flags="--no-cpu-baseline"; fi
  python3 bench.py --workload cfg2 --entities $n --steps $steps --warmup 5 $flags > $out/cfg2_$n.json 2> $out/cfg2_$n.err
done
python3 - <<'PY'
import json, glob
rows = []
for f in sorted(glob.glob("gpurun_out/sweep/cfg2_*.json"), key=lambda p: int(p.split("_")[-1].split(".")[0])):
    lines = [l for l in open(f) if l.startswith("{")]
    if not lines:
        continue
    d = json.loads(lines[-1])
    r, c = d["roofline"], d.get("cpu_baseline")
    rows.append((d["config"]["entities_per_gpu"], d["ms_per_step"], d["ms_per_step_median"], d["value"], r["avg_launch_ms"] * 1e3, r["frac"],
                 r["frac_of_measured_peak"], d["parity"]["visible_set_bit_identical"] and d["parity"]["baked_model_bit_identical"], d["config"]["visible_fraction"],
                 d.get("value_with_block_bounds"), c))
with open("gpurun_out/sweep/r05_size_sweep.txt", "w") as fo:
    fo.write("# bench.py --workload cfg2 --entities N (flat, frustum-only cull + compaction, 1 x MI355X; parity vs the oracle on all N), the CPU\n")
    fo.write("# baseline of the SAME run beside it: AVX2+FMA 8-wide SoA cull of the whole pool (oracle/gv_oracle_avx2.c, bit-identical to the scalar\n")
    fo.write("# restatement of mesh.cpp:111-184), one thread / the fastest thread count around the CPUs the process may use (`cores`)\n")
    cpu = next((r[10] for r in rows if r[10]), None)
    if cpu:
        fo.write(f"# host: {cpu['cpu_model']}, {cpu['nproc']} logical CPUs, {cpu['hardware_threads_allowed']} allowed, cgroup quota {cpu['cgroup_cpu_quota']}; " + open("gpurun_out/sweep/box.txt").read().strip() + "\n")
    fo.write("entities  ms/frame  (median)  culls/s  cull_kernel_us  frac_of_8TB/s  frac_of_box_stream_peak  parity  visible  culls/s_with_block_bounds  cpu_avx2_1t  cpu_avx2_all  cores  threads  cpu_scalar_all  gpu/cpu_all\n")
    for r in rows:
        c = r[10]
        cpu_cols = (f"{c['avx2_soa_cull_1_thread_culls_per_s']:.3e}  {c['value']:.3e}  {c['cores']}  {c['threads_used']['cull']}  {c['scalar_aos_cull_all_threads_culls_per_s']:.3e}  {r[3] / c['value']:.0f}"
                    if c else "-  -  -  -  -  -  (no CPU baseline: not enough host memory for the 10^8 pools + SoA copy)")
        fo.write(f"{r[0]:>9}  {r[1]:.4f}  {r[2]:.4f}  {r[3]:.3e}  {r[4]:.1f}  {r[5]:.3f}  {r[6]:.3f}  {r[7]}  {r[8]:.3f}  {('%.3e' % r[9]) if r[9] else '-'}  {cpu_cols}\n")
print(open("gpurun_out/sweep/r05_size_sweep.txt").read())
PY
