#!/bin/bash
# rocprofv3 counter passes for the encoder's deformable-attention sampler on realistic inputs (tools/gpu_msda_case.py --eager):
# is the kernel bound by the texture-addresser / L1 gather path (TA busy, TCP hit rate), by L2 (TCC hit rate), or by VALU?
# One counter group per pass (--kernel-trace only, as MI355X_MICROARCH.md prescribes); counters that this rocprofv3 does not
# know are dropped from the groups (names differ between releases), the list it does know is kept in counters_available.txt.
tag=${1:-r03}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_msda_$tag
mkdir -p $out
cd /tmp
rocprofv3 -L > $out/counters_available.txt 2>&1
want_groups=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"
  "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
  "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
  "FETCH_SIZE"
  "GRBM_GUI_ACTIVE GRBM_COUNT"
)
i=0
for grp in "${want_groups[@]}"; do
  have=""
  for c in $grp; do
    if grep -qw "$c" $out/counters_available.txt; then have="$have $c"; fi
  done
  i=$((i+1))
  if [ -z "$have" ]; then echo "pass $i: none of [$grp] available"; continue; fi
  timeout 300 rocprofv3 --kernel-trace --pmc $have --output-format csv -d $out/pass$i -- python $GRAFT_REPO_ROOT/tools/gpu_msda_case.py --eager --reps 6 > $out/pass$i.log 2>&1
  echo "pass $i [$have] rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_msda_summary.py $out | tee $out/summary.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete
