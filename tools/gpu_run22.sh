#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "semantic" -s 2>&1 | grep -v Warning | tail -15
echo "=== config 3: K=1203 LVIS vocab, top-300"
timeout 600 python tools/gpu_time_full.py --k 1203 --topk 300 --iters 3 2>&1 | grep -v "^  vit\|backbone" | tail -9
echo "=== config 5: 1536^2, masks + sseg on, top-500"
timeout 600 python tools/gpu_time_full.py --size L_D_1536 --k 133 --semantic --iters 3 2>&1 | tail -6
