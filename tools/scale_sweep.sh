#!/bin/bash
# One command for the multi-GPU curve: bench.py at N = 1, 2, 4, 8 x every exchange pattern, one JSON line
# each into $OUT (default gpurun_out/scale_sweep.jsonl), then a table: value, ms per step, the same ranks without the exchange
# (n1_same_workload), scaling_efficiency, exchange_ms, shard bytes, the bit-shard variant, parity on every rank.
#
#   tools/scale_sweep.sh                                   # weak scaling, cfg5 shape (12.5 M entities per GPU), on the node's GPUs
#   SCALING=strong TOTAL=100000000 tools/scale_sweep.sh    # one 10^8 world cut into N tiles
#   GV_BENCH_BACKEND=gloo GPUS="1 2 8" ENTITIES=200000 tools/scale_sweep.sh   # functional run on a 1-GPU box (never a measurement)
#
# bench.py --gpus N starts its own ranks (fresh children, before anything touches the GPU).
set -u
cd "$(dirname "$0")/.."
GPUS=${GPUS:-"1 2 4 8"}
MODES=${MODES:-"allgather p2p broadcast"}
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
print(f"{'N':>2} {'pattern':>9} {'scaling':>6} {'culls/s':>10} {'ms/step':>8} {'no-exch ms':>10} {'eff(in-line)':>12} {'vs N=1 run':>10} "
      f"{'exch ms':>8} {'shard MB/rank (max)':>19} {'mask ms/step':>12} {'parity ranks':>12}")
for d in rows:
    if "error" in d:
        print(f"{d.get('n_gpus', '?'):>2} ERROR {d['error']}")
        continue
    c, n = d["config"], d["n_gpus"]
    ne = c.get("same_frames_without_exchange") or {}
    mv = c.get("mask_variant") or {}
    b = base.get(d["scaling"])
    # weak: value(N) / (N * value(1)); strong: value(N) / value(1) / N as well (value counts the whole world per step)
    vs1 = d["value"] / (n * b) if b else None
    par = d.get("parity") or {}
    ok = par.get("visible_set_bit_identical") and par.get("baked_model_bit_identical") and par.get("is_visible_identical")
    shard = max(c["shard_bytes_per_rank"]) / 1e6 if c.get("shard_bytes_per_rank") else None
    f = lambda x, spec: format(x, spec) if x is not None else "-"
    print(f"{n:>2} {c.get('exchange_mode') or '-':>9} {d['scaling']:>6} {d['value']:>10.3e} {d['ms_per_step']:>8.4f} {f(ne.get('ms_per_step'), '10.4f'):>10} "
          f"{f(d.get('scaling_efficiency'), '12.3f'):>12} {f(vs1, '10.3f'):>10} {f(c.get('exchange_ms'), '8.3f'):>8} {f(shard, '19.2f'):>19} "
          f"{f(mv.get('ms_per_step'), '12.4f'):>12} {(str(par.get('checked_ranks')) + (' ok' if ok else ' FAIL')):>12}")
EOF
