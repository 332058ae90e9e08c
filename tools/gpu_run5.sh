#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/run5_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/run5_smoke.log; tail -5 gpurun_out/run5_smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 --no-graph --no-cpu-baseline > gpurun_out/run5_bench_eager.log 2>&1; echo "rc=$?" >> gpurun_out/run5_bench_eager.log; tail -3 gpurun_out/run5_bench_eager.log
timeout 900 python bench.py --steps 30 --warmup 3 > gpurun_out/run5_bench_graph.log 2>&1; echo "rc=$?" >> gpurun_out/run5_bench_graph.log; tail -5 gpurun_out/run5_bench_graph.log
