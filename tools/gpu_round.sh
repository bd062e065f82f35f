#!/bin/bash
# GPU box, one call: what a round re-checks after a change of the host side (bench.py, the shim, the test drivers).
#   tools/gpu_round.sh [contract|tier|evidence]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06
what=${1:-contract}
if [ "$what" = contract ]; then
  python -m pytest tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06/bench_contract_tests.txt
  python -m pytest tests/test_headless_tick.py -x -q -m gpu -k "cfg5_in_its_shape or native_exchange" 2>&1 | tail -5 | tee gpurun_out/r06/cfg5_shape_oracle_test.txt
  bash tools/a5_red_check.sh > gpurun_out/r06/a5_red_check.txt 2>&1; grep -E "^==|exit code|\"why\"" gpurun_out/r06/a5_red_check.txt | cut -c1-260
elif [ "$what" = tier ]; then
  python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r06/gpu_tests.txt
  python3 bench.py > gpurun_out/r06/default_bench_line.json 2> gpurun_out/r06/default.err; tail -c 300 gpurun_out/r06/default.err; cut -c1-600 gpurun_out/r06/default_bench_line.json
  bash tools/tick_ranks.sh peers > /dev/null 2>&1; bash tools/tick_ranks.sh communicator --communicator > /dev/null 2>&1
  grep -E "^##|prepare us" gpurun_out/tick_ranks_peers.txt gpurun_out/tick_ranks_communicator.txt | cut -c1-200
  {
    echo "# tests/cpp/headless_tick --mode gpu <args>, GV_TICK_BREAKDOWN=1 (host us per tick of the drop-in's prepare phase), one context"
    for a in "--entities 2000 --ticks 2000" "--entities 10000 --ticks 2000" "--entities 10000 --span-records --ticks 2000" "--entities 100000 --ticks 1000" "--entities 100000 --span-records --ticks 1000" \
             "--entities 10000 --mixed --ticks 2000" "--entities 10000 --mixed --csm --ticks 2000" "--entities 10000 --hier --world --animate 50 --itemised --ticks 2000" \
             "--entities 100000 --mixed --ticks 500" "--entities 300000 --mixed --csm --ticks 200" "--entities 1000000 --ticks 200" "--entities 1000000 --mixed --csm --ticks 200"; do
      echo "## $a"
      GV_TICK_BREAKDOWN=1 timeout 300 ./tests/cpp/build/headless_tick --mode gpu $a 2>&1 | grep -E "prepare us"
    done
  } > gpurun_out/r06/tick.txt 2>&1; cat gpurun_out/r06/tick.txt
elif [ "$what" = evidence ]; then
  # the line an 8-GPU run prints, with 8 ranks SHARING this box's one GPU (torch over gloo, the library's exchange over the tests'
  # shared-memory transport): functional, never a measurement — the probe of the travel patterns, the choice, parity on all ranks
  GV_BENCH_BACKEND=gloo timeout 1500 python3 bench.py --gpus 8 --entities 1500000 --steps 10 --warmup 2 > gpurun_out/r06/gloo8_sample_line.json 2> gpurun_out/r06/gloo8.err
  python3 - gpurun_out/r06/gloo8_sample_line.json <<'PY' | tee gpurun_out/r06/gloo8_summary.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c = d["config"]
print("8 ranks on ONE GPU (functional): exchange_path", c["exchange_path"], "| transport", c["exchange_transport"])
print("exchange_mode", c["exchange_mode"], "chosen from exchange_mode_probe_ms", c["exchange_mode_probe_ms"])
print("visible_by_rank", d["parity"]["visible_by_rank"], "max/mean %.3f" % c["visible_max_over_mean_by_rank"])
print("shard_bytes_per_rank", c["shard_bytes_per_rank"], "gathered/list bytes %.3f" % c["gathered_over_list_bytes"])
print("mode variants", {k: v.get("ms_per_step") for k, v in (c["exchange_mode_variants"] or {}).items()})
print("parity", {k: d["parity"][k] for k in ("visible_set_bit_identical", "is_visible_identical", "baked_model_bit_identical", "checked_ranks")})
PY
  bash tools/collect_traffic.sh > gpurun_out/r06/collect_traffic.log 2>&1; tail -3 gpurun_out/r06/collect_traffic.log
fi
if [ "$what" = soak ]; then
  bash tools/tick_soak.sh ${2:-8} 2>&1 | tail -12 | tee gpurun_out/r06/tick_soak.txt
  bash tools/exchange_soak.sh ${3:-4} 2>&1 | tail -3 | tee gpurun_out/r06/exchange_soak.txt
  bash tools/stress_round.sh > /dev/null 2>&1; cp gpurun_out/r06_stress_parity.txt gpurun_out/r06/stress_parity.txt; tail -12 gpurun_out/r06/stress_parity.txt
fi
