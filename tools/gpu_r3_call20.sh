#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c20
mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_text_tower.py -q -m gpu -k "attention or text" 2>&1 | grep -v Warning | tail -4 | tee $O/pytest_attn.log
timeout 200 python tools/gpu_probe_attn.py 2>&1 | grep -v Warning | grep "ViT\|decoder" | tee $O/attn_probe.log
timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_default.json; cut -c1-200 $O/bench_default.json
