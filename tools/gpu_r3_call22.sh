#!/bin/bash
# the driver's invocation style (20 timed steps after 3 warm-up steps) on the final commit, without the CPU legs
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c22
mkdir -p $O
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_20_steps.json; cut -c1-220 $O/bench_20_steps.json
