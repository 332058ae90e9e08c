#!/bin/bash
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c18
mkdir -p $O
timeout 300 python tools/gpu_probe_kres.py 2>&1 | grep -v Warning | tail -12 | tee $O/kres_probe_old_layout.log
