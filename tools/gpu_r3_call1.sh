#!/bin/bash
# round 3, GPU call 1: validate what was written on the CPU (fused FFN, attention key order, V^T bound), the teacher-forced
# parity tests + error trace, A/B timings, MSDA counters
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c1
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -s -m gpu -k "ffn_fused or attention or msda" 2>&1 | grep -v Warning | tail -25 > $O/pytest_ops.log; tail -3 $O/pytest_ops.log
timeout 300 python -m pytest tests/test_text_tower.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_text.log
timeout 900 python -m pytest tests/test_teacher_forced.py -x -q -s -m gpu 2>&1 | grep -v Warning > $O/pytest_teacher_forced.log; tail -4 $O/pytest_teacher_forced.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "bf16_pipeline or phrase256" 2>&1 | grep -v Warning > $O/pytest_bf16.log; tail -6 $O/pytest_bf16.log
timeout 400 python tools/gpu_error_trace.py L_D_coco80 > $O/bf16_error_trace.log 2>&1; tail -14 $O/bf16_error_trace.log
timeout 200 python tools/gpu_probe_ffn.py 2>&1 | tail -6 | tee $O/ffn_probe.log
timeout 200 python tools/gpu_probe_attn.py 2>&1 | tail -8 | tee $O/attn_probe_perm.log
APE_ATTN_NATURAL_KEY_ORDER=1 timeout 200 python tools/gpu_probe_attn.py 2>&1 | tail -8 | tee $O/attn_probe_natural.log
timeout 200 python tools/gpu_msda_case.py --sigma 0.0 0.5 2.0 2>&1 | tail -4 | tee $O/msda_case.log
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_default.json; cut -c1-400 $O/bench_default.json
APE_FFN_FUSED=0 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_two_gemm_ffn.json; cut -c1-200 $O/bench_two_gemm_ffn.json
APE_FFN_FUSED=b64 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $O/bench_ffn_b64.json; cut -c1-200 $O/bench_ffn_b64.json
./tools/gpu_pmc_msda.sh c1 2>&1 | tail -60
du -sh gpurun_out
