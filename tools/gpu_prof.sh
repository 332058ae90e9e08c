#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (no counters): writes gpurun_out/prof_<tag>/
tag=${1:-r1}
mkdir -p gpurun_out
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-graph > $out.log 2>&1
cd $GRAFT_REPO_ROOT
grep -E '^\{' $out.log | tail -1
f=$(find $out -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
head -45 "$f" | cut -c1-200
