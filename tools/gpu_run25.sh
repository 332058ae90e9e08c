#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -3
python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
