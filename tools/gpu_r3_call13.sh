#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/c13
mkdir -p $O
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o a -- python $GRAFT_REPO_ROOT/tools/gpu_attn_case.py > /tmp/pa.log 2>&1
find /tmp/pa -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-160 | head -8 | tee $O/attn_kernels.txt
