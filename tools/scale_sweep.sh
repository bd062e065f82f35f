#!/bin/bash
# One command for the multi-GPU curve: bench.py at N = 1, 2, 4, 8, one JSON line each into $OUT (default
# gpurun_out/scale_sweep.jsonl), then a table: value, ms per step, the same ranks without the exchange (n1_same_workload),
# scaling_efficiency, exchange_ms, bytes on the links over list bytes, how evenly the ranks share the view, the same frames by
# the other travel patterns / through torch.distributed / as bit shards (all timed inside the one run), parity on every rank.
# The timed exchange is the library's own C-ABI step (gv_exchange_visible; bench.py --exchange-path c-abi, the default).
# MODES="allgather p2p broadcast" runs every pattern as a headline of its own as well.
#
#   tools/scale_sweep.sh                                   # weak scaling, cfg5 shape (12.5 M entities per GPU), on the node's GPUs
#   SCALING=strong TOTAL=100000000 tools/scale_sweep.sh    # one 10^8 world cut into N tiles
#   GV_BENCH_BACKEND=gloo GPUS="1 2 8" ENTITIES=200000 tools/scale_sweep.sh   # functional run on a 1-GPU box (never a measurement)
#
# bench.py --gpus N starts its own ranks (fresh children, before anything touches the GPU).
set -u
cd "$(dirname "$0")/.."
GPUS=${GPUS:-"1 2 4 8"}
MODES=${MODES:-"allgather"}
SCALING=${SCALING:-weak}
STEPS=${STEPS:-50}
WARMUP=${WARMUP:-10}
OUT=${OUT:-gpurun_out/scale_sweep.jsonl}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
extra=()
[ -n "${ENTITIES:-}" ] && extra+=(--entities "$ENTITIES")
[ "$SCALING" = strong ] && extra+=(--scaling strong --entities-total "${TOTAL:-100000000}")
for n in $GPUS; do
    modes=$MODES
    [ "$n" = 1 ] && modes=allgather  # no exchange on one GPU: one run
    for m in $modes; do
        echo "--- --gpus $n --exchange $m ${extra[*]:-} $*" >&2
        # the N = 1 run of a weak sweep is the per-GPU workload of the N > 1 runs (cfg5 shape), not bench.py's N = 1 default (cfg3)
        python bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARMUP" --exchange "$m" --workload "${WORKLOAD:-cfg5}" --no-cpu-baseline \
            "${extra[@]}" "$@" 2>>"${OUT%.jsonl}.err" | grep '^{' >> "$OUT" || echo "{\"error\": \"bench.py --gpus $n --exchange $m failed\", \"n_gpus\": $n}" >> "$OUT"
    done
done
python - "$OUT" <<'EOF'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
base = {}
for d in rows:
    if "error" not in d and d["n_gpus"] == 1:
        base[d["scaling"]] = d["value"]
print(f"{'N':>2} {'path':>5} {'pattern':>9} {'scaling':>6} {'culls/s':>10} {'ms/step':>8} {'no-exch ms':>10} {'eff(in-line)':>12} {'vs N=1 run':>10} "
      f"{'exch ms':>8} {'links/lists':>11} {'vis max/mean':>12} {'other patterns ms':>22} {'torch ms':>8} {'mask ms':>8} {'parity ranks':>12}")
for d in rows:
    if "error" in d:
        print(f"{d.get('n_gpus', '?'):>2} ERROR {d['error']}")
        continue
    c, n = d["config"], d["n_gpus"]
    ne = c.get("same_frames_without_exchange") or {}
    mv = c.get("mask_variant") or {}
    tv = c.get("torch_variant") or {}
    others = " ".join(f"{k}:{v['ms_per_step']:.4f}" if "ms_per_step" in v else f"{k}:ERR" for k, v in (c.get("exchange_mode_variants") or {}).items())
    b = base.get(d["scaling"])
    # weak: value(N) / (N * value(1)); strong: value(N) / value(1) / N as well (value counts the whole world per step)
    vs1 = d["value"] / (n * b) if b else None
    par = d.get("parity") or {}
    ok = par.get("visible_set_bit_identical") and par.get("baked_model_bit_identical") and par.get("is_visible_identical")
    f = lambda x, spec: format(x, spec) if x is not None else "-"
    print(f"{n:>2} {c.get('exchange_path') or '-':>5} {c.get('exchange_mode') or '-':>9} {d['scaling']:>6} {d['value']:>10.3e} {d['ms_per_step']:>8.4f} "
          f"{f(ne.get('ms_per_step'), '10.4f'):>10} {f(d.get('scaling_efficiency'), '12.3f'):>12} {f(vs1, '10.3f'):>10} {f(c.get('exchange_ms'), '8.3f'):>8} "
          f"{f(c.get('gathered_over_list_bytes'), '11.3f'):>11} {f(c.get('visible_max_over_mean_by_rank'), '12.3f'):>12} {others or '-':>22} "
          f"{f(tv.get('ms_per_step'), '8.4f'):>8} {f(mv.get('ms_per_step'), '8.4f'):>8} {(str(par.get('checked_ranks')) + (' ok' if ok else ' FAIL')):>12}")
EOF
