#!/bin/bash
# round-3 closing validation after the store-layout changes (kres / fused FFN / attention): full GPU suite, smoke, the default bench
# line (cpu_baseline + parity + box AP), two more BASELINE configurations, rocprofv3 kernel stats, PMC passes -> gpurun_out/$TAG/
TAG=${1:-final_r3c}
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -4 | tee $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4 | tee $O/smoke.log
timeout 500 python bench.py 2>&1 | tail -1 > $O/bench_default.json; cut -c1-260 $O/bench_default.json
timeout 150 python bench.py --no-cpu-baseline --classes 1203 --size L_D 2>&1 | tail -1 > $O/bench_lvis1203_top300.json; cut -c1-160 $O/bench_lvis1203_top300.json
timeout 150 python bench.py --no-cpu-baseline --stream coco 2>&1 | tail -1 > $O/bench_stream_coco.json; cut -c1-160 $O/bench_stream_coco.json
./tools/gpu_profile.sh $TAG 2>&1 | tail -3 | cut -c1-160
mv gpurun_out/${TAG}_* $O/ 2>/dev/null
rm -f $O/*kernel_trace.csv.gz
./tools/gpu_pmc.sh $TAG 2 2>&1 | tail -14 | cut -c1-220
cp gpurun_out/pmc_$TAG/summary.txt $O/pmc_summary.txt 2>/dev/null
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
du -sh gpurun_out
