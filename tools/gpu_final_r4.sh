#!/bin/bash
# round-4 validation: full GPU suite (regression pins asserted, measured values written), smoke, default bench (cpu_baseline + parity of
# both 16-bit flavours + box AP), the other BASELINE configurations / flavours, rocprofv3 kernel stats (instrumented pass alone + the
# pipelined graph run), PMC passes -> gpurun_out/$TAG/
TAG=${1:-final_r4}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/$TAG
mkdir -p $O
if [ "$2" != "slim" ] && [ "$2" != "nosuite" ]; then
APE_WRITE_PINS=$O timeout 1700 python -m pytest tests -q -m gpu -s 2>&1 | grep -v Warning > $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 | tee $O/smoke.log
fi
timeout 900 python bench.py 2>&1 | tail -1 > $O/bench_default.json; cut -c1-260 $O/bench_default.json
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 > $O/bench_$name.json; cut -c1-170 $O/bench_$name.json; }
b 20_steps --steps 20 --warmup 3
b f16 --dtype f16
b input_uint8 --input uint8
b lvis1203_top300 --classes 1203 --size L_D
b stream_coco --stream coco
b 1536_semantic --size L_D_1536 --semantic --steps 30
b one_image_per_step --images-per-step 1
b L_A --size L_A
b E_D --size E_D --steps 20 --warmup 3
b V_A --size V_A --steps 20 --warmup 3
b V_A_1536 --size V_A_1536 --steps 10 --warmup 2
b G_A --size G_A --steps 10 --warmup 2
if [ "$2" != "slim" ]; then
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_torchrun_n1.json; cut -c1-160 $O/bench_torchrun_n1.json
fi
./tools/gpu_profile.sh $TAG 2>&1 | tail -3 | cut -c1-160
mv gpurun_out/${TAG}_* $O/ 2>/dev/null
rm -f $O/*kernel_trace.csv.gz
./tools/gpu_pmc.sh $TAG 2 2>&1 | tail -14 | cut -c1-220
cp gpurun_out/pmc_$TAG/summary.txt $O/pmc_summary.txt 2>/dev/null
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
du -sh gpurun_out
