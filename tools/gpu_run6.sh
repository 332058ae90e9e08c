#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "gemm" > gpurun_out/run6_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/run6_pytest.log
grep -E "passed|failed|Error|assert|relerr [0-9.e+-]*$" gpurun_out/run6_pytest.log | awk '{ if ($NF+0 > 5e-3 || /passed|failed|Error|assert/) print }' | tail -30
for mode in "APE_GEMM_V1=1" "APE_GEMM_NOGLDS=1" "APE_X=0"; do
  echo "== $mode"; env $mode timeout 600 python tools/gpu_probe.py --out gpurun_out/probe6_${mode%%=*}.json 2>&1 | grep -E "^(vit_|enc_|dec_|mask|big)" 
done
