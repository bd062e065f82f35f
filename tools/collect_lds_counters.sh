#!/bin/bash
# GPU box: SQ counters of the fused sweep + cull kernels (cfg4), current build vs a build with round 1's gv_sweep.hip
# (garden_amd/lib/ab_r01sweep.so, made by hand for this A/B: `git show <round-1 commit>:garden_amd/csrc/gv_sweep.hip`, compiled with the
# Makefile flags and linked with the current objects in place of gv_sweep.o). Counter passes are separate runs with --kernel-trace only.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02e
rm -rf $out; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -iE "SQ_INSTS_LDS|SQ_WAIT_INST_LDS|SQ_INSTS_VALU|MFMA|SQ_ACTIVE_INST_LDS|SQ_LDS_BANK|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES" | head -40 > $out/avail.txt
for lib in cur r01; do
  if [ $lib = r01 ]; then export GV_LIB_PATH=$GRAFT_REPO_ROOT/garden_amd/lib/ab_r01sweep.so; else unset GV_LIB_PATH; fi
  for sweep in fused fused-valu; do
    for grp in "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
      tag=${lib}_${sweep}_$(echo $grp | tr ' ' '+')
      rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$tag -- python3 bench.py --workload cfg4 --sweep $sweep --no-cpu-baseline --no-parity --steps 5 --warmup 2 > $out/$tag.log 2>&1
    done
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = "gpurun_out/r02e"
rows = []
for d in sorted(glob.glob(out + "/*/")):
    tag = os.path.basename(d.rstrip("/"))
    for f in glob.glob(d + "*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "sweep_cull" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            rows.append((tag.split("_SQ")[0], k, c, sum(v) / len(v), len(v)))
with open(out + "/lds_counters.csv", "w") as fo:
    fo.write("build_sweep,kernel,counter,mean_per_launch,launches\n")
    for r in rows:
        fo.write(f'{r[0]},"{r[1]}",{r[2]},{r[3]:.0f},{r[4]}\n')
print(open(out + "/lds_counters.csv").read())
PY
