#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "model_gpu or pipeline or forward_api or 2736 or K40" > gpurun_out/run3_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/run3_pytest.log
grep -E "^\[|passed|failed|Error|error|assert" gpurun_out/run3_pytest.log | tail -60
timeout 900 python tools/gpu_time_full.py --iters 4 > gpurun_out/run3_time.log 2>&1
echo "time rc=$?" >> gpurun_out/run3_time.log
tail -20 gpurun_out/run3_time.log
