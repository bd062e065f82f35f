#!/bin/bash
# Functional checks of the multi-rank bench path on a 1-GPU box (never a measurement of scaling):
#  1. the RCCL code path with a 1-rank group (ExternalStream + all_gather_into_tensor + header checks)
#  2. two ranks sharing the GPU through gloo (host-staged exchange)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "--- N=1 no exchange"; python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | cut -c1-330
echo "--- N=1 rccl 1-rank exchange"; GV_BENCH_EXCHANGE=1 python bench.py --no-cpu-baseline > gpurun_out/ex1.out 2> gpurun_out/ex1.err; echo rc=$?; grep '^{' gpurun_out/ex1.out | cut -c1-1000; tail -5 gpurun_out/ex1.err
echo "--- 2 ranks gloo, 1M each"; GV_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 2 --entities 1000000 2>/dev/null | grep '^{' | cut -c1-400
