#!/bin/bash
# ONE parametrised runner for the GPU calls of a round (replaces the per-call scripts of rounds 3 / 4):
#     gpurun --timeout T -- 'tools/gpu_call.sh <tag> <recipe> [<recipe> ...]'
# Every recipe writes under gpurun_out/<tag>/ (scratch; what is to be judged is copied into profiles/ afterwards).  Recipes:
#   sweep          bench at 2 / 4 / 8 images per step (no CPU baseline): throughput + the tile kernel's roofline by shape
#   meter          the instrumented pass of the default bench under rocprofv3 --kernel-trace --stats: the launch meter's per-kernel
#                  durations on the JSON line next to rocprofv3's kernel_stats.csv of the SAME launches
#   ops            tests/test_ops_gpu.py (+ -k expression in $APE_K)
#   model          tests/test_model_gpu.py tests/test_teacher_forced.py (+ -k expression in $APE_K)
#   suite          the whole -m gpu suite with durations, measured regression values written to the tag directory
#   pytest         python -m pytest $APE_PYTEST_ARGS -m gpu (+ -k expression in $APE_K), measured pins written to the tag directory
#   smoke          __graft_entry__.smoke()
#   bench          the driver's default bench (cpu_baseline + parity) -> bench_default.json
#   benches        the other configurations / flavours (no CPU baseline)
#   profile        tools/gpu_profile.sh (rocprofv3 kernel stats: instrumented pass alone + pipelined graph run)
#   pmc            tools/gpu_pmc.sh (separate FETCH_SIZE / WRITE_SIZE / SQ passes) + summary
#   env            environment variables of the box that steer the HIP / HSA runtimes, clocks
#   d2h            tools/gpu_d2h_probe.py per environment variant under rocprofv3 (is the mask transfer a blit kernel or SDMA?)
#   drivercmd      the driver's exact command (--gpus 1 --steps 20 --warmup 5) with the per-step trace printed
#   py:<script>    python tools/<script> (a probe), output -> <script>.log
TAG=$1; shift
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/$TAG
mkdir -p $O
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" 2> $O/bench_$name.err | tail -1 > $O/bench_$name.json; cut -c1-200 $O/bench_$name.json; tail -2 $O/bench_$name.err | cut -c1-200; }
for recipe in "$@"; do
  echo "=== $recipe"
  case $recipe in
    sweep)
      for B in ${APE_SWEEP:-2 4 8}; do b ips$B --images-per-step $B --steps $((60 / B)) --warmup 4; done
      python tools/bench_digest.py $O/bench_ips*.json | tee $O/sweep_digest.txt ;;
    ablate)
      # marginal cost of whole stages INSIDE the pipelined step (what a stage costs when it overlaps the rest): the step with fewer layers
      b abl_full --steps 30 --warmup 4 --no-second-flavour
      for a in ${APE_ABLATE:-dec_layers=1 enc_layers=1 depth=12}; do b abl_$a --steps 30 --warmup 4 --no-second-flavour --ablate $a; done
      python - <<PY | tee $O/ablate_digest.txt
import json, glob
for f in sorted(glob.glob("$O/bench_abl_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f.split("bench_abl_")[1][:-5].ljust(16), round(d["value"], 2), "images/s", round(d["ms_per_step"], 3), "ms/step")
    except Exception as e:
        print(f, "failed", e)
PY
      ;;
    meter)
      (cd /tmp && rm -rf /tmp/prof_m && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o eager -- python $GRAFT_REPO_ROOT/bench.py --instrumented-only --no-cpu-baseline ${APE_BENCH_ARGS} > $GRAFT_REPO_ROOT/$O/bench_instrumented_under_rocprof.json 2> /tmp/prof_m.err; tail -2 /tmp/prof_m.err)
      find /tmp/prof_m -name "*kernel_stats.csv" -exec cp {} $O/instrumented_kernel_stats.csv \;
      python tools/bench_digest.py --vs-rocprof $O/instrumented_kernel_stats.csv $O/bench_instrumented_under_rocprof.json | tee $O/meter_vs_rocprof.txt ;;
    ops)   timeout 900 python -m pytest tests/test_ops_gpu.py -q -s -m gpu -x ${APE_K:+-k "$APE_K"} 2>&1 | grep -v Warning > $O/pytest_ops.log; tail -4 $O/pytest_ops.log | cut -c1-300 ;;
    model) timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_teacher_forced.py -q -s -m gpu -x ${APE_K:+-k "$APE_K"} 2>&1 | grep -v Warning > $O/pytest_model.log; tail -4 $O/pytest_model.log | cut -c1-300 ;;
    suite) APE_WRITE_PINS=$O timeout 1700 python -m pytest tests -q -s -m gpu --durations=40 2>&1 | grep -v Warning > $O/pytest_gpu.log; tail -50 $O/pytest_gpu.log | cut -c1-200 ;;
    pytest) APE_WRITE_PINS=$O timeout ${APE_PYTEST_TIMEOUT:-1700} python -m pytest ${APE_PYTEST_ARGS:-tests} -q -s -m gpu ${APE_K:+-k "$APE_K"} --durations=15 2>&1 | grep -v Warning > $O/pytest.log; grep -E "passed|failed|error" $O/pytest.log | tail -5 | cut -c1-300; grep -E "^FAILED|^ERROR|assert|Error" $O/pytest.log | head -20 | cut -c1-300 ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 | tee $O/smoke.log ;;
    bench) timeout 900 python bench.py ${APE_BENCH_ARGS} 2> $O/bench_default.err | tail -1 > $O/bench_default.json; cut -c1-300 $O/bench_default.json; python tools/bench_digest.py $O/bench_default.json | tee $O/bench_default_digest.txt ;;
    benches)
      b f16 --dtype f16; b input_uint8 --input uint8; b lvis1203_top300 --classes 1203 --size L_D; b stream_coco --stream coco
      b 1536_semantic --size L_D_1536 --semantic --steps 30; b one_image_per_step --images-per-step 1; b L_A --size L_A
      timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_torchrun_n1.json; cut -c1-160 $O/bench_torchrun_n1.json
      # the N > 1 code path end to end on a 1-GPU box: two ranks SHARE the device (gloo: RCCL refuses two ranks on one GPU) -- mask format
      # "both", lagged all-gather of records + run lengths, per-rank rates, the same-process N = 1 reference and efficiency_vs_n1.  The
      # throughput of such a run means nothing; that every field is produced does
      APE_BENCH_SHARE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --backend gloo --steps 10 --warmup 2 --solo-steps 6 --no-cpu-baseline 2> $O/bench_n2_shared_gpu_gloo.err | tail -1 > $O/bench_n2_shared_gpu_gloo.json; cut -c1-200 $O/bench_n2_shared_gpu_gloo.json; tail -2 $O/bench_n2_shared_gpu_gloo.err | cut -c1-200 ;;
    benches_wide)   # the widened model families: APE on ViT-e, EVA-01 MIM ViT-g (1024^2 / 1536^2), EVA-01-CLIP ViT-g
      b E_D --size E_D --steps 20 --warmup 3 --no-second-flavour; b V_A --size V_A --steps 20 --warmup 3 --no-second-flavour
      b V_A_1536 --size V_A_1536 --steps 10 --warmup 2 --no-second-flavour; b G_A --size G_A --steps 10 --warmup 2 --no-second-flavour ;;
    profile) ./tools/gpu_profile.sh $TAG ${APE_BENCH_ARGS} 2>&1 | tail -3 | cut -c1-160; mv gpurun_out/${TAG}_* $O/ 2>/dev/null; rm -f $O/*kernel_trace.csv.gz ;;
    pmc) ./tools/gpu_pmc.sh $TAG ${APE_PMC_GROUPS:-2} 2>&1 | tail -14 | cut -c1-220; cp gpurun_out/pmc_$TAG/summary.txt $O/pmc_summary.txt 2>/dev/null ;;
    env) env | grep -E "^(HSA|HIP|ROC|GPU|AMD|NCCL|RCCL|PYTORCH)" | sort | tee $O/env.txt; rocm-smi --showclocks --showpower 2>/dev/null | head -30 | tee -a $O/env.txt ;;
    d2h)   # the mask transfer: blit kernel or SDMA?  one probe run per environment variant, each under rocprofv3 --kernel-trace --stats
      i=0
      for v in ${APE_D2H_VARIANTS:-X=0 HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0 HSA_ENABLE_SDMA=0 HSA_ENABLE_SDMA_COPY_SIZE_OVERRIDE=0}; do
        i=$((i + 1))
        (cd /tmp && rm -rf /tmp/prof_d$i && env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d$i -o p -- python $GRAFT_REPO_ROOT/tools/gpu_d2h_probe.py > $GRAFT_REPO_ROOT/$O/d2h_$i.log 2>&1)
        echo "--- variant $v" | tee -a $O/d2h_summary.txt
        grep -E "^env|copy alone|sdma copy correct|Error|error" $O/d2h_$i.log | tee -a $O/d2h_summary.txt
        f=$(find /tmp/prof_d$i -name "*kernel_stats.csv" | head -1)
        (grep -i -E "copyBuffer|blit|fill" "$f" | cut -d, -f1-4 || echo "no copy kernel in the trace") | tee -a $O/d2h_summary.txt
      done ;;
    drivercmd)   # EXACTLY the driver's command, ${APE_REPEAT:-2} times back to back on this box, with the per-step trace on the line
      for r in $(seq 1 ${APE_REPEAT:-2}); do
        timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ${APE_BENCH_ARGS} 2> $O/drivercmd_$r.err | tail -1 > $O/drivercmd_$r.json
        python - <<PY | tee -a $O/drivercmd_trace.txt
import json
d = json.loads(open("$O/drivercmd_$r.json").read())
print("run $r: value", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 3), "f16", round(d.get("value_f16") or 0, 2), "frac", round(d["roofline"]["frac"], 4), "avg_us", round(d["roofline"]["avg_launch_us"], 2))
t = d.get("step_trace") or {}
print("  warmup_ms", t.get("warmup_ms")); print("  timed_ms ", t.get("timed_ms"))
t = d.get("step_trace_f16") or {}
print("  f16 warmup_ms", t.get("warmup_ms")); print("  f16 timed_ms ", t.get("timed_ms"))
PY
      done ;;
    py:*) s=${recipe#py:}; timeout 900 python tools/$s ${APE_PY_ARGS} > $O/${s%.py}.log 2>&1; tail -40 $O/${s%.py}.log | cut -c1-220 ;;
    *) echo "unknown recipe $recipe" ;;
  esac
done
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
du -sh gpurun_out | tail -1
