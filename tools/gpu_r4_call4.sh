#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c4; mkdir -p $O
timeout 200 python tools/gpu_gemm_p8.py 8192x2048x1024 16384x2048x1024 4096x2048x1024 2>&1 | grep -v Warning > $O/gemm_p8_table.log; cat $O/gemm_p8_table.log | cut -c1-220
timeout 300 python bench.py --no-cpu-baseline --steps 50 2>&1 | tail -1 > $O/bench_bf16.json; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r4c4/bench_bf16.json').read()); ro=r['roofline']
print(r['value'], ro['frac'], ro['avg_launch_us']); print(ro['by_shape'])
PY
