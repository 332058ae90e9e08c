#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "gemm" 2>&1 | tail -3
python tools/gpu_probe_small.py 2>&1 | grep -E "^M|floor"
