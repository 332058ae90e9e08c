#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c21
mkdir -p $O
timeout 300 python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "small_G" 2>&1 | grep -v Warning | grep -E "small_G\] (pred_logits|p2 |detections|vit_blk3|pred_boxes)|passed|failed|rror" | tail -14 | tee $O/pytest_small_G.log
