#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu -s -p no:cacheprovider > gpurun_out/run4_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/run4_pytest.log
grep -E "^\[|passed|failed|Error|error|assert" gpurun_out/run4_pytest.log | tail -40
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -- python $GRAFT_REPO_ROOT/tools/gpu_time_full.py --iters 6 > $GRAFT_REPO_ROOT/gpurun_out/run4_time.log 2>&1
cd $GRAFT_REPO_ROOT
tail -12 gpurun_out/run4_time.log
find gpurun_out/prof4 -name "*kernel_stats*" | head
f=$(find gpurun_out/prof4 -name "*kernel_stats.csv" | head -1); head -40 "$f"
