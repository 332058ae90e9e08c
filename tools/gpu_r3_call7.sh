#!/bin/bash
# round 3, GPU call 7: ViT-e (HD = 128 attention, post-norm residual kernel, backbone + small full model), control AP
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c7
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or postnorm" 2>&1 | grep -v Warning | tail -6 | tee $O/pytest_ops.log
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -s -m gpu -k "vite or small_E or jpeg or photograph" 2>&1 | grep -v Warning | grep -E "ViT-e|passed|failed|Error|error|jpeg" | tail -30 | tee $O/pytest_model.log
timeout 300 python tools/gpu_probe_vite.py 2>&1 | grep -v Warning | tail -4 | tee $O/vite_probe.log
timeout 600 python tools/gpu_ap_control.py 2>&1 | grep -v Warning | tail -3 | tee $O/ap_control.log
