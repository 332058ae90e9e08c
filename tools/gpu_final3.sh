#!/bin/bash
# round-end validation (round 2, second half): full GPU suite, smoke, default bench (with cpu_baseline + parity), the other
# BASELINE configurations, rocprofv3 kernel stats (isolated eager pass + the pipelined graph run), PMC passes -> gpurun_out/final3/
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/final3
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -4 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4 | tee $O/smoke.log
timeout 600 python bench.py 2>&1 | tail -1 > $O/bench_default.json; cut -c1-260 $O/bench_default.json
timeout 200 python bench.py --no-cpu-baseline --classes 1203 --size L_D 2>&1 | tail -1 > $O/bench_lvis1203_top300.json; cut -c1-160 $O/bench_lvis1203_top300.json
timeout 200 python bench.py --no-cpu-baseline --stream coco 2>&1 | tail -1 > $O/bench_stream_coco.json; cut -c1-160 $O/bench_stream_coco.json
timeout 200 python bench.py --no-cpu-baseline --images-per-step 1 2>&1 | tail -1 > $O/bench_one_image_per_step.json; cut -c1-160 $O/bench_one_image_per_step.json
./tools/gpu_profile.sh final3 2>&1 | tail -3 | cut -c1-160
mv gpurun_out/final3_* $O/ 2>/dev/null
rm -f $O/*kernel_trace.csv.gz
./tools/gpu_pmc.sh final3 2 2>&1 | tail -14 | cut -c1-220
cp gpurun_out/pmc_final3/summary.txt $O/pmc_summary.txt 2>/dev/null
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
du -sh gpurun_out
