#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c3; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "gemm" 2>&1 | grep -v Warning | tail -15 > $O/pytest_gemm.log; tail -6 $O/pytest_gemm.log | cut -c1-250
timeout 200 python tools/gpu_gemm_p8.py 8192x2048x1024 8192x5504x1024 16384x2048x1024 8192x1024x2752 65536x256x2304 2>&1 | grep -v Warning > $O/gemm_p8_table.log; cat $O/gemm_p8_table.log | cut -c1-220
timeout 300 python bench.py --no-cpu-baseline --steps 50 2>&1 | tail -1 > $O/bench_bf16.json; cut -c1-200 $O/bench_bf16.json
timeout 300 python bench.py --no-cpu-baseline --steps 50 --input uint8 2>&1 | tail -1 > $O/bench_bf16_uint8.json; cut -c1-200 $O/bench_bf16_uint8.json
