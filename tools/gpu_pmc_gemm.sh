#!/bin/bash
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
mkdir -p $out; cd /tmp
run() { tag=$1; shift; grp=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$tag -- python $GRAFT_REPO_ROOT/tools/gpu_one_gemm.py "$@" > $out/$tag.log 2>&1; echo "$tag rc=$?"; }
for shape in "4096 2048 1024 rope" "4096 5504 1024 swiglu" "87296 2048 256 relu"; do
  t=$(echo $shape | tr ' ' '_')
  run ${t}_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" $shape
  run ${t}_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" $shape
  run ${t}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU" $shape
done
cd $GRAFT_REPO_ROOT
for d in $out/*/; do echo "== $d"; python tools/pmc_summary.py $d | grep -i "gemm" | cut -c1-260; done
