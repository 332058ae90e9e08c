#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c14
mkdir -p $O
for a in 0 16 32 48; do
  echo "--- variant $a (16 = s_setprio 1 in the matrix phase, 32 = DMA issued in the matrix phase)" | tee -a $O/abl2.log
  APE_ATTN_PP_ABL=$a timeout 100 python tools/gpu_probe_attn.py 2>&1 | grep -v Warning | grep "ViT global  \|ViT windowed  " | tee -a $O/abl2.log
done
APE_ATTN_PP_ABL=48 timeout 120 python -m pytest tests/test_ops_gpu.py -q -s -m gpu -k "variants_agree" 2>&1 | grep -v Warning | grep -E "passed|failed|rror" | tail -3 | tee -a $O/abl2.log
