#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "dense_fusion or phrase" -s 2>&1 | grep -v Warning | tail -40
timeout 600 python tools/gpu_time_full.py --prompt phrase --k 20 --iters 3 2>&1 | tail -12
