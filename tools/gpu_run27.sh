#!/bin/bash
export TMPDIR=/tmp
python bench.py --steps 30 --warmup 3 2>&1 | tail -1 > gpurun_out/bench_r1f.json; cut -c1-1800 gpurun_out/bench_r1f.json
./tools/gpu_prof.sh r1f 2>&1 | sed -n 2,14p | cut -c1-150
./tools/gpu_pmc.sh r1f 2>&1 | tail -30 | cut -c1-220
