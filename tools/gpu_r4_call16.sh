#!/bin/bash
# round 4, call 16: implicit-GEMM 3x3 convolution (bit-identity vs im2col + gemm, timing), one full-size pipeline case, bench
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/call16
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -s -x -k "conv3x3_implicit or (L_D_bf16_pipeline and L_D_coco80)" 2>&1 | grep -v Warning > $O/pytest.log; tail -3 $O/pytest.log | cut -c1-300
grep -n "conv3x3\|implicit" $O/pytest.log | cut -c1-200 | head -20
timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>&1 | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json
