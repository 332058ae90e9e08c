#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c7; mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline --steps 50 2>&1 | tail -1 > $O/bench_bf16.json; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r4c7/bench_bf16.json').read()); ro=r['roofline']
print(r['value'], ro['frac'], ro['avg_launch_us'], ro['metering'][:80]); print(ro['by_shape']); g=ro['all_gemm_kernels']; print(g['ms_per_image'], g['tflops']); print(g['by_kernel_ms_per_image']); print(g['by_kernel_tflops'])
PY
timeout 300 python bench.py --no-cpu-baseline --steps 50 --dtype f16 2>&1 | tail -1 > $O/bench_f16.json; cut -c1-120 $O/bench_f16.json
