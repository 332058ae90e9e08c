#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c11
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -s -m gpu -k "four_query" 2>&1 | grep -v Warning | grep -E "QT=4|passed|failed|rror" | tail -8 | tee $O/pytest_qt4.log
echo "--- default" | tee $O/attn_probe.log
timeout 300 python tools/gpu_probe_attn.py 2>&1 | grep -v Warning | grep "ViT" | tee -a $O/attn_probe.log
echo "--- APE_ATTN_QT4=1" | tee -a $O/attn_probe.log
APE_ATTN_QT4=1 timeout 300 python tools/gpu_probe_attn.py 2>&1 | grep -v Warning | grep "ViT" | tee -a $O/attn_probe.log
timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_default.json; cut -c1-200 $O/bench_default.json
APE_ATTN_QT4=1 timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_qt4.json; cut -c1-200 $O/bench_qt4.json
