#!/bin/bash
# GPU session 1: per-op parity + micro-benchmarks. Logs go to gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/run1_device.txt
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider > gpurun_out/run1_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/run1_pytest.log
tail -40 gpurun_out/run1_pytest.log
timeout 600 python tools/gpu_probe.py --out gpurun_out/probe1.json > gpurun_out/run1_probe.log 2>&1
echo "probe rc=$?" >> gpurun_out/run1_probe.log
tail -40 gpurun_out/run1_probe.log
