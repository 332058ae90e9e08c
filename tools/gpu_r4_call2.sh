#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c2; mkdir -p $O
timeout 300 python tools/gpu_p8_ablate.py 2>&1 | grep -v Warning > $O/p8_ablation.log; cat $O/p8_ablation.log | cut -c1-200
