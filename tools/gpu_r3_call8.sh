#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c8
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -s -m gpu -k "attention or postnorm" 2>&1 | grep -v Warning | grep -E "attention|passed|failed|Error" | tail -40 | tee $O/pytest_ops.log
timeout 600 python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "vite" 2>&1 | grep -v Warning | tail -8 | tee $O/pytest_vite.log
timeout 600 python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "test_bf16_pipeline and small_E" 2>&1 | grep -v Warning | grep -E "small_E\]|passed|failed" | head -70 | tee $O/pytest_small_E.log
