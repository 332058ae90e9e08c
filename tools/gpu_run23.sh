#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm" 2>&1 | grep -v Warning | tail -8
timeout 300 python tools/gpu_probe_kres.py 2>&1 | grep -v Warn | tail -14
