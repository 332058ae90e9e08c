#!/bin/bash
# round 4, call 14: the opt-in f16 teacher-forced cases (APE_TEST_ALL_F16=1) and V_A at full size, measured values written as pins
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/call14
mkdir -p $O
APE_WRITE_PINS=$O APE_TEST_ALL_F16=1 timeout 1000 python -m pytest tests/test_teacher_forced.py -q -m gpu -s -k "1536 or L_A or jpeg or E_D or V_A" 2>&1 | grep -v Warning > $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
grep -n "head choice flipped\|autocast" $O/pytest.log | cut -c1-330 | tail -20
