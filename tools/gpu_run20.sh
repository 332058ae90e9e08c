#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-600
