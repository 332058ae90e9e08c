#!/bin/bash
# round 3, GPU call 3: fused FFN with the two-groups-ahead prefetch, the plain (L_A) family, store-flavour experiment, TA counters
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c3
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -s -m gpu -k "ffn_fused or attention or msda" 2>&1 | grep -v Warning | tail -30 > $O/pytest_ops.log; tail -2 $O/pytest_ops.log
timeout 200 python tools/gpu_probe_ffn.py 2>&1 | tail -2 | tee $O/ffn_probe.log
timeout 300 python tools/gpu_p8_store_probe.py 2>&1 | tail -8 | tee $O/p8_store_probe.log
timeout 900 python -m pytest tests/test_teacher_forced.py -q -s -m gpu 2>&1 | grep -v Warning > $O/pytest_teacher_forced.log; tail -3 $O/pytest_teacher_forced.log; grep -h "EXCEEDS\|flipped" $O/pytest_teacher_forced.log | cut -c1-200
timeout 900 python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "small_A or L_A or phrase256" 2>&1 | grep -v Warning > $O/pytest_plain_family.log; tail -4 $O/pytest_plain_family.log; grep -h "^\[L_D" $O/pytest_plain_family.log | grep "L_A" | cut -c1-260
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_default.json; cut -c1-200 $O/bench_default.json
timeout 300 python bench.py --no-cpu-baseline --size L_A 2>&1 | tail -1 > $O/bench_L_A.json; cut -c1-200 $O/bench_L_A.json
cd /tmp
for grp in "TA_BUSY_avr TA_ADDR_STALL_CYCLES_sum"; do
  have=""
  for c in $grp; do if grep -qw "$c" $GRAFT_REPO_ROOT/gpurun_out/pmc_msda_c1/counters_available.txt 2>/dev/null || rocprofv3 -L 2>/dev/null | grep -qw "$c"; then have="$have $c"; fi; done
  [ -z "$have" ] && { echo "no TA counter of [$grp]"; continue; }
  name=$(echo $have | tr ' ' '+' | cut -c1-30)
  timeout 90 rocprofv3 --kernel-trace --pmc $have --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$name -- python $GRAFT_REPO_ROOT/tools/gpu_msda_case.py --eager --reps 2 > $GRAFT_REPO_ROOT/$O/pmc_$name.log 2>&1
  echo "TA pass [$have] rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_msda_summary.py $O 2>&1 | tail -12 | tee $O/ta_summary.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh gpurun_out
