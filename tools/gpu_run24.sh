#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -5
timeout 300 python tools/gpu_probe_kres.py 2>&1 | grep -v Warn | tail -12
