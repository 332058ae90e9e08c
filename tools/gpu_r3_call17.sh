#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c17
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "gemm or rope or swiglu or conv or ffn" 2>&1 | grep -v Warning | tail -4 | tee $O/pytest_gemm.log
timeout 300 python tools/gpu_gemm_table.py 2>&1 | grep -v Warning | tail -16 | tee $O/p8_table.log
timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_default.json; cut -c1-200 $O/bench_default.json
