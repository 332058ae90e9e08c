#!/bin/bash
# round 4, call 11: proposal_order as chunk sorts + rank merge (bit-exact selection tests, kernel time under rocprofv3), and the
# teacher-forced L_D_coco80 cases against the re-measured pins
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/call11
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_teacher_forced.py tests/test_model_gpu.py -q -m gpu -s -k "select_proposals or two_stage or (teacher_forced and L_D_coco80) or any_size_runtime" 2>&1 | grep -v Warning > $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
cd /tmp; rm -rf /tmp/prof_po
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_po -o po -- python $GRAFT_REPO_ROOT/bench.py --instrumented-only --no-cpu-baseline > /dev/null 2> /tmp/po.err
find /tmp/prof_po -name "*kernel_stats.csv" -exec grep -h "proposal_order\|proposal_topk2\|topk_stage1\|nms_scan\|proposal_quota" {} \; | cut -c1-200 | tee $O/selection_kernel_stats.csv
