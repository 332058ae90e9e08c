#!/bin/bash
# round 3, GPU call 4: fused FFN (reads / VALU inside the MFMA stream, LayerNorm epilogue), top-k merge level, text-tower changes
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c4
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -s -m gpu -k "ffn_fused or detections or attention or msda" 2>&1 | grep -v Warning | tail -40 > $O/pytest_ops.log; tail -3 $O/pytest_ops.log; grep -h "ffn_fused" $O/pytest_ops.log | tail -8
timeout 300 python -m pytest tests/test_text_tower.py -x -q -m gpu 2>&1 | tail -2 | tee $O/pytest_text.log
timeout 200 python tools/gpu_probe_ffn.py 2>&1 | tail -2 | tee $O/ffn_probe.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "fp32_pipeline or test_bf16_pipeline or graph or runtime" 2>&1 | grep -v Warning | tail -4 | tee $O/pytest_model_small.log
timeout 600 python -m pytest tests/test_teacher_forced.py -q -s -m gpu -k "coco80" 2>&1 | grep -v Warning > $O/pytest_teacher_forced.log; tail -2 $O/pytest_teacher_forced.log; grep -h "EXCEEDS\|enc5_out\|enc0_out" $O/pytest_teacher_forced.log | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_default.json; cut -c1-200 $O/bench_default.json
APE_FFN_LN=0 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_no_ln_fusion.json; cut -c1-120 $O/bench_no_ln_fusion.json
du -sh gpurun_out
