#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c9
mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_teacher_forced.py -q -m gpu -k "small_E or maskprompt or jpeg or photograph or vite" 2>&1 | grep -v Warning | tail -12 | tee $O/pytest_new.log
