#!/bin/bash
# Register / LDS / spill table of every kernel of the library, from the compiler's own notes (hipcc -S, device side only):
#   tools/regs.sh > profiles/rNN_register_usage.txt
# What to look for: sgpr_spill / vgpr_spill > 0 (SGPR spills live in VGPR lanes: a v_readlane per use; VGPR spills go to scratch),
# and VGPR counts just above a multiple of 8 x (512 / waves) — 64 VGPRs = 8 waves per SIMD, 72 = 7, 80 = 6.
cd "$(dirname "$0")/../garden_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
printf "%-72s %5s %5s %6s %6s %8s %7s\n" kernel vgpr sgpr s_spl v_spl scratch lds
for f in gv_cull gv_sweep gv_hiz gv_sort gv_reorder gv_shard gv_probe; do
  /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o /tmp/regs_$f.s $f.hip 2>/dev/null || { echo "$f: compile failed"; continue; }
  python3 - /tmp/regs_$f.s <<'PY'
import re, subprocess, sys
text = open(sys.argv[1]).read()
for block in re.findall(r"- \.agpr_count:.*?\.wavefront_size:", text, re.S):
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", block)
    name = g("name").group(1)
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    print("%-72s %5s %5s %6s %6s %8s %7s" % (name[:72], g("vgpr_count").group(1), g("sgpr_count").group(1), g("sgpr_spill_count").group(1),
                                            g("vgpr_spill_count").group(1), g("private_segment_fixed_size").group(1), g("group_segment_fixed_size").group(1)))
PY
done
