#!/usr/bin/env python3
"""Post-processes tools/collect_traffic.sh output (gpurun_out/pmc) into profiles/: traffic.json (HBM bytes per
launch of the cull kernel from the FETCH_SIZE / WRITE_SIZE passes) and per-workload kernel stats / bench lines.
  python tools/summarize_traffic.py r01f
The read side is calibrated on the kernel's own known stream (see _method in traffic.json)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC = os.path.join(ROOT, "gpurun_out", "pmc")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01x"


def newest(pattern):
    files = glob.glob(os.path.join(PMC, pattern))
    return max(files, key=os.path.getmtime) if files else None


def counter_mean(workload, counter, kernel_prefix):
    f = newest(f"{workload}_{counter}/*/*counter_collection.csv")
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and kernel_prefix in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


N = 10_000_000
sys.path.insert(0, ROOT)
import datetime  # noqa: E402

from garden_amd.benchlib.workloads import kernel_source_sha  # noqa: E402  (bench.py reports traffic only while this hash still matches)

out = {"_kernel_source_sha": kernel_source_sha(ROOT), "_collected": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ") + f" ({tag})",
       "_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/collect_traffic.sh, "
       "tools/summarize_traffic.py). gfx950 FETCH_SIZE under-reports wide coalesced reads (MI355X_MICROARCH.md: 1/2 for "
       "16 B/lane; other widths uncalibrated), so the read side is calibrated on this kernel's own access pattern: the "
       "frustum-only cull of a flat, exactly-paired pool of 10 M entities streams a known 65 B/entity (+ one 8-byte flag "
       "word per wave) -> factor = known bytes / (FETCH_SIZE KB * 1024); WRITE_SIZE taken as reported (KB * 1024). Per "
       "launch of gv::cull_kernel. The factor is applied to all reads of cfg3 too, which over-counts its 8-byte Hi-Z "
       "texel gathers (narrow reads are reported closer to 1:1): cfg3's figure is an upper bound."}
PLAIN, BOUNDED = "gv::cull_kernel<", "gv::cull_list_kernel"  # cull_kernel<HIZ, MAP>; the bounded path's per-entity kernel (the kept-block list form)
f2, n2 = counter_mean("cfg2", "FETCH_SIZE", PLAIN)
w2, _ = counter_mean("cfg2", "WRITE_SIZE", PLAIN)
known = N * 65 + N / 64 * 8
factor = known / (f2 * 1024)
out["fetch_calibration_factor"] = factor
out["cfg2_at_10M"] = {"entities": N, "FETCH_SIZE_KB": f2, "WRITE_SIZE_KB": w2, "launches": n2,
                      "cull_kernel_hbm_bytes_per_launch": f2 * 1024 * factor + w2 * 1024}
f3, n3 = counter_mean("cfg3", "FETCH_SIZE", PLAIN)
w3, _ = counter_mean("cfg3", "WRITE_SIZE", PLAIN)
out["cfg3"] = {"entities": N, "FETCH_SIZE_KB": f3, "WRITE_SIZE_KB": w3, "launches": n3,
               "cull_kernel_hbm_bytes_per_launch": f3 * 1024 * factor + w3 * 1024,
               "cull_kernel_hbm_bytes_per_launch_uncalibrated": f3 * 1024 + w3 * 1024}
if newest("cfg3hard_FETCH_SIZE/*/*counter_collection.csv"):  # cfg3 on scene.noise_depth (bench.py --depth noise): the queries reach levels 0-2
    fh, nh = counter_mean("cfg3hard", "FETCH_SIZE", PLAIN)
    wh, _ = counter_mean("cfg3hard", "WRITE_SIZE", PLAIN)
    out["cfg3_hard_depth"] = {"entities": N, "FETCH_SIZE_KB": fh, "WRITE_SIZE_KB": wh, "launches": nh,
                              "cull_kernel_hbm_bytes_per_launch": fh * 1024 * factor + wh * 1024,
                              "cull_kernel_hbm_bytes_per_launch_uncalibrated": fh * 1024 + wh * 1024}
if newest("cfg3bb_FETCH_SIZE/*/*counter_collection.csv"):
    fb, nb = counter_mean("cfg3bb", "FETCH_SIZE", BOUNDED)
    wb, _ = counter_mean("cfg3bb", "WRITE_SIZE", BOUNDED)
    out["cfg3_block_bounds"] = {"FETCH_SIZE_KB": fb, "WRITE_SIZE_KB": wb, "launches": nb,
                                "kernel": "gv::cull_list_kernel (+ block_classify_kernel, block_window_kernel: a few hundred KB each)",
                                "cull_kernel_hbm_bytes_per_launch": fb * 1024 * factor + wb * 1024}
if newest("cfg4_FETCH_SIZE/*/*counter_collection.csv"):
    f4, n4 = counter_mean("cfg4", "FETCH_SIZE", "sweep_cull_mfma_kernel")
    w4, _ = counter_mean("cfg4", "WRITE_SIZE", "sweep_cull_mfma_kernel")
    out["cfg4"] = {"entities": N, "FETCH_SIZE_KB": f4, "WRITE_SIZE_KB": w4, "launches": n4, "kernel": "gv::sweep_cull_mfma_kernel",
                   "cull_kernel_hbm_bytes_per_launch": f4 * 1024 * factor + w4 * 1024}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.csv"), "w") as fo:
    fo.write("workload,kernel,counter,mean_value_KB,launches\n")
    for wl in ("cfg2", "cfg3", "cfg3hard", "cfg3bb", "cfg4"):
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            f = newest(f"{wl}_{counter}/*/*counter_collection.csv")
            if not f:
                continue
            acc = {}
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            for k, v in sorted(acc.items()):
                if k.startswith("gv::") or "gv::" in k:
                    fo.write(f'{wl}{"@10M" if wl == "cfg2" else ""},"{k}",{counter},{sum(v) / len(v):.3f},{len(v)}\n')

for wl in ("cfg2", "cfg3", "cfg3hard", "cfg4", "cfg3bb", "cfg4valu", "cfg2_10M", "cfg5shape", "driver", "default"):
    ks = newest(f"stats_{wl}/*/*kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(ROOT, "profiles", f"{tag}_{wl}_kernel_stats.csv"))
    log = os.path.join(PMC, f"stats_{wl}.log")
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            open(os.path.join(ROOT, "profiles", f"{tag}_{wl}_bench_line.json"), "w").write(lines[-1])
