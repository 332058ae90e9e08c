#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -s -k "small_padded" 2>&1 | grep -vE "^\s*$" | tail -60
