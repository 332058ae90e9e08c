#!/bin/bash
# round 4, call 15: the N = 2 code path of bench.py on ONE GPU (two ranks sharing it, gloo for the collectives): rle masks + records
# all-gathered with lag 1, text bank broadcast, barrier / max-over-ranks timing -- a functional smoke, not a scaling number
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 APE_BENCH_SHARE_GPU=1
O=gpurun_out/call15
mkdir -p $O
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --backend gloo --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_n2_shared_gpu_gloo.log 2>&1
tail -3 $O/bench_n2_shared_gpu_gloo.log | cut -c1-1200
