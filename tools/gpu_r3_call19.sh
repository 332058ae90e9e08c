#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c19
mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -q -s -m gpu -k "ffn" 2>&1 | grep -v Warning | tail -12 | tee $O/pytest_ffn.log
timeout 200 python tools/gpu_probe_ffn.py 2>&1 | tail -3 | tee $O/ffn_probe.log
timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_default.json; cut -c1-200 $O/bench_default.json
