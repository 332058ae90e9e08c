#!/bin/bash
# round 3, GPU call 2: explicit fragment prefetch in the fused FFN, half values in the sampler, attention without the mid-tile DMA drain
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c2
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -s -m gpu -k "ffn_fused or attention or msda" 2>&1 | grep -v Warning | tail -40 > $O/pytest_ops.log; tail -3 $O/pytest_ops.log
timeout 300 python -m pytest tests/test_text_tower.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_text.log
timeout 200 python tools/gpu_probe_ffn.py 2>&1 | tail -5 | tee $O/ffn_probe.log
timeout 200 python tools/gpu_probe_attn.py 2>&1 | tail -6 | tee $O/attn_probe.log
APE_ATTN_OCC4=1 timeout 200 python tools/gpu_probe_attn.py 2>&1 | tail -6 | tee $O/attn_probe_occ4.log
timeout 200 python tools/gpu_msda_case.py --check --sigma 0.5 2>&1 | tail -3 | tee $O/msda_case.log
timeout 900 python -m pytest tests/test_teacher_forced.py -q -s -m gpu 2>&1 | grep -v Warning > $O/pytest_teacher_forced.log; tail -4 $O/pytest_teacher_forced.log; grep -h "EXCEEDS\|flipped" $O/pytest_teacher_forced.log | cut -c1-220
timeout 900 python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "bf16_pipeline or phrase256 or coco80" 2>&1 | grep -v Warning > $O/pytest_bf16.log; tail -6 $O/pytest_bf16.log
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_default.json; cut -c1-200 $O/bench_default.json
APE_MSDA_BF16_VALUE=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_bf16_value.json; cut -c1-120 $O/bench_bf16_value.json
APE_FFN_FUSED=0 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_two_gemm_ffn.json; cut -c1-120 $O/bench_two_gemm_ffn.json
du -sh gpurun_out
