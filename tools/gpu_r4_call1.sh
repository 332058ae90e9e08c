#!/bin/bash
# round-4 call 1: the whole GPU suite with the f16 flavour (writes the measured stage errors for the regression pins),
# default bench in both flavours, today's p8 GEMM table
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c1; mkdir -p $O
APE_WRITE_PINS=$O timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -v Warning > $O/pytest_gpu.log; tail -25 $O/pytest_gpu.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --steps 50 2>&1 | tail -1 > $O/bench_bf16.json; cut -c1-200 $O/bench_bf16.json
timeout 300 python bench.py --no-cpu-baseline --steps 50 --dtype f16 2>&1 | tail -1 > $O/bench_f16.json; cut -c1-200 $O/bench_f16.json
timeout 200 python tools/gpu_gemm_p8.py 2>&1 | grep -v Warning > $O/gemm_p8_table.log; tail -20 $O/gemm_p8_table.log | cut -c1-220
