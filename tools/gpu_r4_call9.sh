#!/bin/bash
# round 4, call 9: APE-E_D at full size -- the chaos-aware fp32 comparison with the reference run + the teacher-forced bf16 stages
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/call9
mkdir -p $O
APE_WRITE_PINS=$O timeout 900 python -m pytest tests/test_model_gpu.py tests/test_teacher_forced.py -q -m gpu -s -x -k "E_D" 2>&1 | grep -v Warning > $O/pytest.log; tail -5 $O/pytest.log | cut -c1-300
grep -n "E_D" $O/pytest.log | cut -c1-260 | head -90
