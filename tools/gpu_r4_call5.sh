#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c5; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "kres or patchify or ffn_fused" 2>&1 | grep -v Warning | tail -6 > $O/pytest_ops.log; tail -4 $O/pytest_ops.log | cut -c1-250
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "L_D_coco80 or small or tiny" -s 2>&1 | grep -v Warning > $O/pytest_model.log; tail -5 $O/pytest_model.log | cut -c1-250
timeout 300 python bench.py --no-cpu-baseline --steps 50 2>&1 | tail -1 > $O/bench_bf16.json; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r4c5/bench_bf16.json').read()); ro=r['roofline']
print(r['value'], ro['frac'], ro['avg_launch_us']); print(ro['by_shape']); print(ro['all_gemm_kernels']['by_kernel_ms_per_image'])
PY
