#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c15
mkdir -p $O
timeout 120 python -m pytest tests/test_ops_gpu.py -q -s -m gpu -k "variants_agree" 2>&1 | grep -v Warning | grep -E "variants|passed|failed|rror" | tail -12 | tee $O/pytest_sw.log
echo "--- APE_ATTN_SW=0" | tee $O/attn_probe.log
APE_ATTN_SW=0 timeout 120 python tools/gpu_probe_attn.py 2>&1 | grep -v Warning | grep "ViT" | tee -a $O/attn_probe.log
echo "--- software-pipelined (default)" | tee -a $O/attn_probe.log
timeout 120 python tools/gpu_probe_attn.py 2>&1 | grep -v Warning | grep "ViT" | tee -a $O/attn_probe.log
