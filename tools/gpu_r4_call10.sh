#!/bin/bash
# round 4, call 10: panoptic merge on the device (op test, forward() vs the reference fixture, in the graph runtime) + the semantic
# branch in the size-agnostic graph
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/call10
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -s -k "size_agnostic or any_size" 2>&1 | grep -v Warning > $O/pytest.log; tail -25 $O/pytest.log | cut -c1-300
