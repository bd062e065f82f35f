#!/bin/bash
# GPU box: the round's text evidence (sort probe, tick breakdown, default bench line) -> gpurun_out/evidence
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/evidence
rm -rf $out; mkdir -p $out
# a probe is (re)built from the sources of THIS snapshot or not run at all: no binary left in /tmp by an earlier call is ever
# measured in its place, and the compiler's output is kept beside the evidence
build_probe() {  # build_probe <binary> <hipcc arguments...>
    local bin=$1; shift
    rm -f "$bin"
    if ! hipcc "$@" -o "$bin" 2>>"$out/hipcc_stderr.txt"; then
        echo "FAILED: hipcc $* (see hipcc_stderr.txt): $bin was not built, its measurements are missing below"
        return 1
    fi
}
run_probe() {  # run_probe <binary> <arguments...>: only a binary build_probe has just made
    [ -x "$1" ] && timeout 120 "$@" || echo "SKIPPED: $* (not built)"
}
{
  echo "# tools/onesweep_probe: gv_sort (rank kernel + scatter kernel per digit), capacity 10 M slots; times in us"
  build_probe /tmp/onesweep_plain --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Igarden_amd/csrc tools/onesweep_probe.hip
  build_probe /tmp/onesweep_probe --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DGV_SORT_TRACE -Igarden_amd/csrc tools/onesweep_probe.hip
  run_probe /tmp/onesweep_plain 2124723 10000000
  run_probe /tmp/onesweep_plain 308383 10000000
  run_probe /tmp/onesweep_plain 9900000 10000000
  echo "# the same with wall-clock stamps per tile and phase (-DGV_SORT_TRACE; a few us slower)"
  run_probe /tmp/onesweep_probe 2124723 10000000
  echo
  echo "# tools/sort_bench.py (through the C-ABI, cull + emit + gv_sort per frame, 50 frames; hipEvents)"
  timeout 200 python3 tools/sort_bench.py 2>&1 | grep records
  echo
  echo "# tools/permute_probe: the last pass's job in isolation — 48-byte records through a random permutation"
  build_probe /tmp/permute_probe --offload-arch=gfx950 -O3 tools/permute_probe.hip
  run_probe /tmp/permute_probe 2124723
  echo
  echo "# history of the large sort (2 124 723 records, same probe / same sizes):"
  echo "#   round 1: 14 launches (4 x hist / scan / scatter, keys, gather)                                   349 us"
  echo "#   round 2a: onesweep — one launch per digit with a decoupled look-back (+ a histogram launch)      223-231 us; a pass = 38-42 us of which 9-16 us"
  echo "#             waiting for predecessor tiles (every tile is resident at once: the look-back is a serial chain through memory); a 64-word window: 269 us"
  echo "#   round 2b (kept): rank kernel + scatter kernel per digit, no inter-workgroup waiting at all      203-207 us; a pass = 11 + 2 + 10 us, the last one + 90 us"
  echo "#             of record gather (the isolated gather above: 70 us — sector-granular random reads, the floor of this pass)"
  echo "#   round 2c (kept): the same two kernels with 512-lane workgroups, and 1024-key tiles for lists of up to 512 k records (chosen on the device)   190-195 us; 308 k records 100 -> 62 us, 21.7 k 71 -> 42 us"
  echo "#   round 2d (kept): pool slots carried beside the pairs (the last pass gathers only the models) and a leaner digit match in the rank kernel      175-182 us"
  echo "#   ranking variants measured on the way (kept: 8-ballot match + wave-private LDS running counts): LDS lane-mask tables (no gain), 4 rotating mask tables"
  echo "#             with returning LDS atomics (spills at 128 VGPRs), static tile ids under the look-back form (-12 us, unsafe there; the kept form needs no ids)"
} > $out/sort_probe.txt 2>&1
{
  echo "# tests/cpp/headless_tick --mode gpu --ticks 2000 <args>, GV_TICK_BREAKDOWN=1 (host us per tick of the drop-in's prepare phase)"
  for a in "--entities 2000" "--entities 10000" "--entities 10000 --span-records" "--entities 10000 --copy-records" "--entities 100000" "--entities 100000 --span-records" "--entities 100000 --copy-records" "--entities 10000 --mixed" "--entities 10000 --mixed --span-records" "--entities 10000 --mixed --csm" "--entities 10000 --hier --world --animate 50 --itemised"; do
    echo "## $a"
    GV_TICK_BREAKDOWN=1 ./tests/cpp/build/headless_tick --mode gpu --ticks 2000 $a 2>&1 | grep -E "prepare us"
  done
  echo "# --span-records (round 4): the render passes read the library's page-locked result buffer (UnsortedBuffer::meshes()); nothing is copied into combinedMeshes"
  echo "# round 1 (profiles/r01k_tick_*): 2 k 32-35, 10 k 49-52, 100 k 263, 10 k --mixed 210, --mixed --csm 203; round 2 (profiles/r02_tick.txt, records into page-locked engine vectors): 2 k 21-22, 10 k 33.8-36, 100 k 161-173, --mixed --csm 76-86; round 3 (profiles/r03_tick.txt, engine vectors never page-locked): 100 k 216"
} > $out/r06_tick.txt 2>&1
{
  echo "# cfg3 on the HARD depth image (bench.py --depth noise: per-8x8-block occluders among the entities), then on the walls (SURVEY.md 8d)"
  for d in "--depth noise" ""; do
    echo "## bench.py $d"
    python3 bench.py $d --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f  cull kernel %.4f ms  frac %.3f  visible %.4f  frame kernels %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['visible_fraction'], d['config']['frame_kernel_ms']))"
  done
} > $out/r06_hard_depth.txt 2>&1
{
  echo "# tools/hiz_sizes.py: pyramid rebuild by frame size, wall clock over 300 back-to-back rebuilds (us)"
  timeout 120 python3 tools/hiz_sizes.py 2>&1 | grep rebuild
} > $out/r06_hiz_sizes.txt 2>&1
{
  echo "# tools/multiview_bench.py: main camera + 3 cascades over 10 M entities, one batched pass vs one pass per view (ms per frame; kernel us per frame)"
  timeout 300 python3 tools/multiview_bench.py 2>&1 | grep -E "batched|separate"
} > $out/r06_multiview.txt 2>&1
# the line an 8-GPU run prints, with 8 ranks SHARING this box's one GPU (torch over gloo, the library's exchange over the tests'
# shared-memory transport): functional, never a measurement — what it shows is the balance of the ranks and the bytes on the links
GV_BENCH_BACKEND=gloo timeout 1500 python3 bench.py --gpus 8 --entities 1500000 --steps 10 --warmup 2 > $out/r06_gloo8_sample_line.json 2> $out/gloo8.err
python3 - $out/r06_gloo8_sample_line.json <<'EOF' > $out/r06_gloo8_summary.txt 2>&1
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c = d["config"]
print("8 ranks on ONE GPU (functional): exchange_path", c["exchange_path"], "| transport", c["exchange_transport"])
print("visible_by_rank", d["parity"]["visible_by_rank"], "max/mean %.3f" % c["visible_max_over_mean_by_rank"])
print("list_bytes_per_rank", c["list_bytes_per_rank"])
print("shard_bytes_per_rank", c["shard_bytes_per_rank"], "gathered/list bytes %.3f" % c["gathered_over_list_bytes"])
print("mode variants", {k: v.get("shard_bytes_per_rank") for k, v in (c["exchange_mode_variants"] or {}).items()})
print("parity", {k: d["parity"][k] for k in ("visible_set_bit_identical", "is_visible_identical", "baked_model_bit_identical", "checked_ranks")})
EOF
python3 bench.py > $out/default_bench_line.json 2> $out/default.err
tail -c 600 $out/default.err
