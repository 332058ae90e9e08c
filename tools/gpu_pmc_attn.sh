#!/bin/bash
# rocprofv3 counter passes for the flash-attention kernel on the two production launches (tools/gpu_attn_case.py): how busy are
# the VALU, the matrix pipe and the LDS, and how much of a wave's life is waiting?  One counter group per pass, --kernel-trace only.
tag=${1:-r03}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn_$tag
mkdir -p $out
cd /tmp
rocprofv3 -L > $out/counters_available.txt 2>&1
want_groups=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_MFMA"
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
  "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INSTS_SALU"
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
  timeout 300 rocprofv3 --kernel-trace --pmc $have --output-format csv -d $out/pass$i -- python $GRAFT_REPO_ROOT/tools/gpu_attn_case.py > $out/pass$i.log 2>&1
  echo "pass $i [$have] rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_msda_summary.py $out | tee $out/summary.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete
rm -f $out/counters_available.txt
