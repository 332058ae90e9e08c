#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c6; mkdir -p $O
timeout 300 python tools/gpu_probe_cold.py 2>&1 | grep -v Warning > $O/cold_probe.log; cat $O/cold_probe.log | cut -c1-300
