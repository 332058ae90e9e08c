#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "gemm" 2>&1 | tail -2
for mode in "APE_GEMM_NORING=1" "APE_X=0"; do
  echo "== $mode"; env $mode timeout 600 python tools/gpu_probe.py --out gpurun_out/probe13_${mode%%=*}.json 2>&1 | grep -E "^(vit_|enc_|dec_|mask|big)" | sed -e "s/'M': //; s/'N': //; s/'K': //"
done
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
