#!/bin/bash
# refresh of the bench lines / kernel stats after the two-stage top-k (the full suite + PMC passes of gpu_final3.sh stay valid)
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/final4
mkdir -p $O
timeout 400 python bench.py 2>&1 | tail -1 > $O/bench_default.json; cut -c1-200 $O/bench_default.json
timeout 150 python bench.py --no-cpu-baseline --classes 1203 --size L_D 2>&1 | tail -1 > $O/bench_lvis1203_top300.json; cut -c1-160 $O/bench_lvis1203_top300.json
timeout 150 python bench.py --no-cpu-baseline --stream coco 2>&1 | tail -1 > $O/bench_stream_coco.json; cut -c1-160 $O/bench_stream_coco.json
timeout 150 python bench.py --no-cpu-baseline --images-per-step 1 2>&1 | tail -1 > $O/bench_one_image_per_step.json; cut -c1-160 $O/bench_one_image_per_step.json
timeout 150 python bench.py --no-cpu-baseline --images-per-step 3 2>&1 | tail -1 > $O/bench_three_images_per_step.json; cut -c1-160 $O/bench_three_images_per_step.json
cd /tmp && rm -rf /tmp/prof_e && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o eager -- env APE_NO_FORK=1 python $GRAFT_REPO_ROOT/bench.py --no-graph --images-per-step 2 --no-pipeline --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_eager_under_rocprof.json 2> /tmp/prof_e.err
find /tmp/prof_e -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/$O/eager_kernel_stats.csv \;
cd $GRAFT_REPO_ROOT; head -12 $O/eager_kernel_stats.csv | cut -c1-150
