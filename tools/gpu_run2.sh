#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "patchify or spatial or nms or vl_pool or mask_post" > gpurun_out/run2_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/run2_pytest.log
tail -30 gpurun_out/run2_pytest.log
