#!/bin/bash
# rocprofv3 PMC passes of the bench command (eager launches, few steps).  One counter group per pass, as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass); --kernel-trace only.
tag=${1:-r1}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp
groups=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES")
ngrp=${2:-3}
for grp in "${groups[@]:0:$ngrp}"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$name -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-flavour --no-graph --no-pipeline --images-per-step 2 > $out/$name.log 2>&1
  echo "pass [$grp] rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out | tee $out/summary.txt
