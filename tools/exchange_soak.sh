#!/bin/bash
# GPU box: tests/cpp/exchange_ranks with cameras that look somewhere else through another lens every frame (--random-camera), 2 .. 8
# ranks sharing the box's GPU over the test transport's DEVICE form (collectives = kernels on the library's exchange stream), every
# travel pattern, 32 frames each: every row of every frame must be its owner's WHOLE list on every rank. -> gpurun_out/exchange_soak.txt
# usage: tools/exchange_soak.sh [seeds per rank count, default 12]
set -u
cd "$GRAFT_REPO_ROOT"
seeds=${1:-12}
out=gpurun_out/exchange_soak.txt
: > $out
export GV_RCCL_LIBRARY=$PWD/tests/cpp/build/librccl_stub.so
runs=0; bad=0; frames=0; second=0; rows=0; tails=0
for ranks in 2 3 4 8; do
  for seed in $(seq 1 $seeds); do
    for mode in all allgather p2p broadcast; do
      # (every other seed: two lists per frame in ONE exchange, gv_exchange_views)
      batched=""; [ $((seed % 2)) = 0 ] && batched="--batched"
      line=$(timeout 300 ./tests/cpp/build/exchange_ranks --ranks $ranks --entities 40000 --frames 32 --mode $mode --random-camera $((seed * 131 + ranks)) $batched 2>>$out.err | tail -1)
      runs=$((runs + 1))
      ok=$(echo "$line" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(int(d['ok'] and d['mismatches']==0), d['frames'], d['frames_with_a_second_exchange'], d['short_rows_completed'], d['tail_words'])" 2>/dev/null || echo "0 0 0 0 0")
      set -- $ok
      [ "$1" = "1" ] || { bad=$((bad + 1)); echo "FAILED ranks $ranks seed $seed mode $mode: $line" >> $out; }
      frames=$((frames + $2)); second=$((second + $3)); rows=$((rows + $4)); tails=$((tails + $5))
    done
  done
done
# the same cameras through the one-process peer path (gv_exchange_init_peers: no communicator; every row of every rank compared word
# for word with its owner's own list)
pruns=0; pbad=0
for ranks in 2 3 4 8; do
  for seed in $(seq 1 $seeds); do
    batched=""; [ $((seed % 2)) = 0 ] && batched="--batched"
    line=$(timeout 300 ./tests/cpp/build/exchange_ranks --peers --ranks $ranks --entities 40000 --frames 32 --random-camera $((seed * 131 + ranks)) $batched 2>>$out.err | tail -1)
    pruns=$((pruns + 1))
    echo "$line" | grep -q '"ok": true' || { pbad=$((pbad + 1)); echo "FAILED peers ranks $ranks seed $seed: $line" >> $out; }
  done
done
echo "exchange soak, peer stores (one process, no communicator): $pruns runs (2 / 3 / 4 / 8 ranks x $seeds random cameras, 32 frames each, every other camera with two lists per frame): $pbad failed" | tee -a $out
echo "exchange soak: $runs runs (2 / 3 / 4 / 8 ranks x $seeds random cameras x 4 travel settings, 32 frames each, 40 000 entities per rank; every other camera with two lists per frame in one exchange): $bad failed; $frames frames, $second of them needed a second exchange ($rows short rows completed, $tails words in tails); every row of every frame == its owner's whole list on every rank" | tee -a $out
