#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -s -k "gemm" 2>&1 | grep -E "splitk|passed|failed|Error" | tail -14
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
