#!/bin/bash
# round-end validation: full GPU suite, smoke, bench (default flags), torchrun N=1, rocprofv3 kernel stats, PMC passes
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -3 | tee gpurun_out/final/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/final/smoke.log
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/final/bench_default.json; cut -c1-300 gpurun_out/final/bench_default.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/final/bench_torchrun.json; cut -c1-200 gpurun_out/final/bench_torchrun.json
./tools/gpu_prof.sh final 2>&1 | sed -n 2,8p | cut -c1-150
./tools/gpu_pmc.sh final 2>&1 | tail -12 | cut -c1-200
