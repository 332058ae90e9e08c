#!/bin/bash
# rocprofv3 kernel-trace + stats of (a) the bench's step composition run eagerly with every branch inline (APE_NO_FORK=1:
# kernels run one at a time, so the per-kernel averages are isolated durations) and (b) the default bench (graph replay,
# software pipeline); summaries -> gpurun_out/
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -x
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_e /tmp/prof_g
# (a) = the instrumented pass of the default bench and nothing else (--instrumented-only): the launches the HIP events of the `roofline`
# object metered, each starting on a busy GPU (spin kernel ahead of every step); the per-kernel averages of this trace are what
# `roofline.avg_launch_us` must agree with
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o eager -- python $R/bench.py --instrumented-only --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_bench_eager_under_rocprof.json 2> /tmp/prof_e.err
find /tmp/prof_e -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_eager_kernel_stats.csv \;
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o graph -- python $R/bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-second-flavour > $R/gpurun_out/${TAG}_bench_graph_under_rocprof.json 2> /tmp/prof_g.err
find /tmp/prof_g -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_graph_kernel_stats.csv \;
find /tmp/prof_g -name "*kernel_trace.csv" -exec python $R/tools/trace_busy.py {} \; > $R/gpurun_out/${TAG}_graph_busy.txt 2>&1
find /tmp/prof_g -name "*kernel_trace.csv" -exec gzip -c {} \; > $R/gpurun_out/${TAG}_graph_kernel_trace.csv.gz
find /tmp/prof_e -name "*kernel_trace.csv" -exec python $R/tools/trace_busy.py {} \; > $R/gpurun_out/${TAG}_eager_busy.txt 2>&1
tail -n 3 /tmp/prof_e.err; tail -n 3 /tmp/prof_g.err; find /tmp/prof_e /tmp/prof_g -type f | head
