#!/bin/bash
# round 4, call 18: the 16-bit full-size model tests + runtime tests + smoke on the code with the implicit 3x3 convolution (results are
# bit-identical to the im2col path, so every regression pin must hold unchanged)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/call18
mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_teacher_forced.py -q -m gpu -x -k "L_D_bf16_pipeline or (teacher_forced and (L_D_coco80 or 1536 or L_A)) or semantic or panoptic or any_size or parallel_images or software_pipelined or rle" 2>&1 | grep -v Warning > $O/pytest.log; tail -3 $O/pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 | tee $O/smoke.log
