#!/bin/bash
# full GPU suite with the measured regression values written out (APE_WRITE_PINS) + smoke
TAG=${1:-suite_r4}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/$TAG
mkdir -p $O
APE_WRITE_PINS=$O timeout 2400 python -m pytest tests -q -m gpu -s --durations=15 2>&1 | grep -v Warning > $O/pytest_gpu.log; tail -30 $O/pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 | tee $O/smoke.log
