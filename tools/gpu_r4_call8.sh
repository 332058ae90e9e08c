#!/bin/bash
# round 4, call 8: the vit_eva.py flavour (wide-key attention, relpos gather, small_V model) + the full-size E_D fixture + V_A bench lines
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/call8
mkdir -p $O
APE_WRITE_PINS=$O timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -s -x -k "wide_keys or relpos_extend or small_V or E_D_coco80" 2>&1 | grep -v Warning > $O/pytest.log; tail -5 $O/pytest.log | cut -c1-300
grep -n "small_V\|E_D\|attention wide\|relpos" $O/pytest.log | cut -c1-260 | tail -60
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 > $O/bench_$name.json; cut -c1-400 $O/bench_$name.json; }
b V_A --size V_A --steps 20 --warmup 3
b V_A_1536 --size V_A_1536 --steps 10 --warmup 2
