#!/bin/bash
# round 4, call 12: kernel profile of the EVA-01 MIM ViT-g flavour (V_A, 1024^2): where the relative-position formulation spends time
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/call12
mkdir -p $O
cd /tmp; rm -rf /tmp/prof_va
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_va -o va -- python $GRAFT_REPO_ROOT/bench.py --size V_A --steps 10 --warmup 2 --no-cpu-baseline --no-graph --no-pipeline > $O/bench_V_A_eager.json 2> /tmp/va.err
find /tmp/prof_va -name "*kernel_stats.csv" -exec cp {} $O/V_A_eager_kernel_stats.csv \;
head -25 $O/V_A_eager_kernel_stats.csv | cut -c1-170
tail -2 /tmp/va.err
