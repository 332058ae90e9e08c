#!/bin/bash
# profile outputs only (small files): default bench line, rocprofv3 kernel stats, PMC summary
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/final/bench_default.json; cut -c1-200 gpurun_out/final/bench_default.json
./tools/gpu_prof.sh final 2>&1 | sed -n 2,4p | cut -c1-150
find gpurun_out -name "*kernel_trace.csv" -delete
./tools/gpu_pmc.sh final 2 2>&1 | tail -3 | cut -c1-200
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
du -sh gpurun_out
