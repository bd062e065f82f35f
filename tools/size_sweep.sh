#!/bin/bash
# GPU box: bench.py --workload cfg2 (flat, frustum-only) at 10^4 .. 10^8 entities on ONE GPU -> gpurun_out/sweep
# (BASELINE north_star: "throughput on synthetic scenes of 10^4-10^8 entities ... as absolute numbers and as fraction of
# the HBM-read roofline")
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sweep
rm -rf $out; mkdir -p $out
for n in 10000 100000 1000000 10000000 100000000; do
  steps=200; [ $n -ge 100000000 ] && steps=20
  python3 bench.py --workload cfg2 --entities $n --steps $steps --warmup 5 --no-cpu-baseline > $out/cfg2_$n.json 2> $out/cfg2_$n.err
done
python3 - <<'PY'
import json, glob
rows = []
for f in sorted(glob.glob("gpurun_out/sweep/cfg2_*.json"), key=lambda p: int(p.split("_")[-1].split(".")[0])):
    d = json.loads(open(f).read())
    r = d["roofline"]
    rows.append((d["config"]["entities_per_gpu"], d["ms_per_step"], d["ms_per_step_median"], d["value"], r["avg_launch_ms"] * 1e3, r["frac"],
                 r["frac_of_measured_peak"], d["parity"]["visible_set_bit_identical"] and d["parity"]["baked_model_bit_identical"], d["config"]["visible_fraction"],
                 d.get("value_with_block_bounds")))
with open("gpurun_out/sweep/r04_size_sweep.txt", "w") as fo:
    fo.write("# bench.py --workload cfg2 --entities N (flat, frustum-only cull + compaction, 1 x MI355X; parity vs the oracle on all N)\n")
    fo.write("entities  ms/frame  (median)  culls/s  cull_kernel_us  frac_of_8TB/s  frac_of_box_stream_peak  parity  visible  culls/s_with_block_bounds\n")
    for r in rows:
        fo.write(f"{r[0]:>9}  {r[1]:.4f}  {r[2]:.4f}  {r[3]:.3e}  {r[4]:.1f}  {r[5]:.3f}  {r[6]:.3f}  {r[7]}  {r[8]:.3f}  {('%.3e' % r[9]) if r[9] else '-'}\n")
print(open("gpurun_out/sweep/r04_size_sweep.txt").read())
PY
