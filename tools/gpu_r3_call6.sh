#!/bin/bash
# round 3, GPU call 6: the glue-removal kernels (unit tests + the model tests that cover the rewired paths) and the bench
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c6
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "language_side or gather_rows or query_init or det_records or gemv or vl_pool or box_refine" 2>&1 | grep -v Warning | tail -8 | tee $O/pytest_ops.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_teacher_forced.py -x -q -m gpu 2>&1 | grep -v Warning | tail -6 | tee $O/pytest_model.log
timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_default.json; cut -c1-200 $O/bench_default.json
timeout 200 python bench.py --no-cpu-baseline --images-per-step 1 2>&1 | tail -1 > $O/bench_b1.json; cut -c1-200 $O/bench_b1.json
