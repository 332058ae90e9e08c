#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "phrase_prompt_through or forward_api" 2>&1 | grep -v Warning | tail -3
python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
