#!/bin/bash
# round-3 validation: full GPU suite, smoke, default bench (cpu_baseline + parity + box AP), the other BASELINE configurations,
# the launch forms (--gpus 1 self-launch, torchrun), rocprofv3 kernel stats (isolated eager pass + pipelined graph run), PMC
# passes, bf16 error trace -> gpurun_out/$TAG/
TAG=${1:-final_r3}
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/$TAG
mkdir -p $O
if [ "$2" != "slim" ]; then
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -4 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4 | tee $O/smoke.log
fi
timeout 700 python bench.py 2>&1 | tail -1 > $O/bench_default.json; cut -c1-260 $O/bench_default.json
timeout 200 python bench.py --no-cpu-baseline --classes 1203 --size L_D 2>&1 | tail -1 > $O/bench_lvis1203_top300.json; cut -c1-160 $O/bench_lvis1203_top300.json
timeout 200 python bench.py --no-cpu-baseline --stream coco 2>&1 | tail -1 > $O/bench_stream_coco.json; cut -c1-160 $O/bench_stream_coco.json
timeout 200 python bench.py --no-cpu-baseline --images-per-step 1 2>&1 | tail -1 > $O/bench_one_image_per_step.json; cut -c1-160 $O/bench_one_image_per_step.json
timeout 200 python bench.py --no-cpu-baseline --size L_A 2>&1 | tail -1 > $O/bench_L_A.json; cut -c1-160 $O/bench_L_A.json
if [ "$2" != "slim" ]; then
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_torchrun_n1.json; cut -c1-160 $O/bench_torchrun_n1.json
timeout 300 python tools/gpu_error_trace.py L_D_coco80 2>&1 | grep -v Warning > $O/bf16_error_trace.log; tail -22 $O/bf16_error_trace.log | cut -c1-200
fi
./tools/gpu_profile.sh $TAG 2>&1 | tail -3 | cut -c1-160
mv gpurun_out/${TAG}_* $O/ 2>/dev/null
rm -f $O/*kernel_trace.csv.gz
./tools/gpu_pmc.sh $TAG 2 2>&1 | tail -14 | cut -c1-220
cp gpurun_out/pmc_$TAG/summary.txt $O/pmc_summary.txt 2>/dev/null
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
du -sh gpurun_out
