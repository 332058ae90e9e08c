"""bf16 error trace of the benchmarked pipeline (APE-L_D, 1024^2, seeded weights of the reference-generated fixture).

Prints, per stage, the TEACHER-FORCED error of the bf16 HIP pipeline (each stage fed the fp32 pipeline's input: the stage's own
error) next to its FREE-RUNNING error (accumulated), then follows the reference boxes through the six decoder layers: where
does the max-norm box error of the bench line's `parity` object come from?  Method and tolerances: tests/teacher_forced.py.

    python tools/gpu_error_trace.py [case ...]  > profiles/r03_bf16_error_trace.log
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import model_util as M  # noqa: E402
import teacher_forced as TF  # noqa: E402


def box_trace(tag, outs):
    t, free, forced = outs["teacher"], outs["free_stages"], outs["forced_stages"]
    nd = max(int(k[3:-4]) for k in t if k.startswith("dec") and k.endswith("_ref")) + 1
    print(f"[{tag}] decoder reference boxes (sigmoid space, absolute error vs the fp32 pipeline), per layer:")
    print(f"[{tag}]   layer | forced: rms      max   | free: rms      max    #q>1e-2  #q>5e-2 | query stream free rms | delta free rms")
    worst = None
    for i in range(nd):
        ef = (forced[f"dec{i}_ref"].float() - t[f"dec{i}_ref"].float()).abs()
        er = (free[f"dec{i}_ref"].float() - t[f"dec{i}_ref"].float()).abs()
        q = er.max(dim=1)[0]
        so = TF.rel_rms(free[f"dec{i}_out"].float(), t[f"dec{i}_out"].float())
        sd = TF.rel_rms(free[f"dec{i}_delta"].float(), t[f"dec{i}_delta"].float())
        print(f"[{tag}]   {i:5d} | {ef.pow(2).mean().sqrt().item():.2e} {ef.max().item():.2e} | {er.pow(2).mean().sqrt().item():.2e} "
              f"{er.max().item():.2e} {int((q > 1e-2).sum()):8d} {int((q > 5e-2).sum()):8d} | {so:.2e}              | {sd:.2e}")
        worst = int(q.argmax())
    e0 = (free["init_reference"].float() - t["init_reference"].float()).abs().max(dim=1)[0]
    print(f"[{tag}] init_reference (sigmoid of the selected proposals' boxes): free max {e0.max().item():.2e}")
    print(f"[{tag}] the query with the largest final box error is #{worst}; its error by layer (free-running, max over the 4 coordinates):")
    traj = [float((free[f'dec{i}_ref'][worst].float() - t[f'dec{i}_ref'][worst].float()).abs().max()) for i in range(nd)]
    strm = [TF.rel_rms(free[f"dec{i}_out"][worst].float(), t[f"dec{i}_out"][worst].float()) for i in range(nd)]
    print(f"[{tag}]   box   " + "  ".join(f"{v:.2e}" for v in traj))
    print(f"[{tag}]   query " + "  ".join(f"{v:.2e}" for v in strm))
    box = t[f"dec{nd - 1}_ref"][worst].float().tolist()
    print(f"[{tag}]   its fp32 box (cx, cy, w, h) = " + ", ".join(f"{v:.4f}" for v in box))
    # distribution of the final error over the queries
    er = (free[f"dec{nd - 1}_ref"].float() - t[f"dec{nd - 1}_ref"].float()).abs().max(dim=1)[0]
    qs = torch.quantile(er.cpu(), torch.tensor([0.5, 0.9, 0.99, 1.0]))
    print(f"[{tag}] final box error over the {er.numel()} queries: median {qs[0]:.2e}  p90 {qs[1]:.2e}  p99 {qs[2]:.2e}  max {qs[3]:.2e}")


def main():
    cases = sys.argv[1:] or ["L_D_coco80"]
    dev = "cuda"
    if os.environ.get("APE_TEST_SELFCHECK") == "1":          # harness check on the CPU: ops := their torch definitions
        import ape_amd.ops as ops
        import ref_ops
        dev = "cpu"
        for n in dir(ref_ops):
            if not n.startswith("_") and callable(getattr(ref_ops, n)) and hasattr(ops, n):
                setattr(ops, n, getattr(ref_ops, n))
    for case in cases:
        model, image, text, gold = M.build_model(case, dev, torch.float32)
        image, text = image.to(dev), text.to(dev)
        ref_topk = gold["full"]["topk_proposals"][0].to(dev)
        mv = model.model_vision
        mv.set_compute_dtype(torch.float32)
        teacher = TF.StageTap()
        mv.forward_single(image, text, forced_topk=ref_topk, stages=teacher)
        mv.set_compute_dtype(torch.bfloat16)
        forced = TF.StageTap(teacher=teacher)
        mv.forward_single(image, text, forced_topk=ref_topk, stages=forced)
        free = TF.StageTap()
        mv.forward_single(image, text, forced_topk=ref_topk, stages=free)
        biases, scales = TF.head_biases(model), TF.linear_head_scales(model, teacher)
        ferr, rerr = TF.stage_errors(forced, teacher, biases, scales), TF.stage_errors(free, teacher, biases, scales)
        TF.report(case, ferr, rerr)
        box_trace(case, dict(teacher=teacher, free_stages=free, forced_stages=forced))
        bad = TF.violations(ferr)
        print(f"[{case}] stages outside their derived tolerance: {sorted(bad) if bad else 'none'}")
        del model, teacher, forced, free
        if dev == "cuda":
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
