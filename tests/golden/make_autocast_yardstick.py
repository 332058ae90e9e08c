"""The reference's OWN code at reduced precision, as a yardstick for the 16-bit HIP pipelines (build container only: needs
/root/reference).  For each case the reference runs in fp32 and again under torch.autocast("cpu", bfloat16) -- with the fp32 run's
proposals injected, so that what differs is arithmetic, not a different selection -- and the relative rms difference of every
captured stage is stored in tests/golden/autocast_yardstick.json.  tests/test_model_gpu.py asserts that the bf16 HIP pipeline is
at least as close to fp32 as this at every stage the two share (and prints the f16 pipeline beside it).

    python tests/golden/make_autocast_yardstick.py [case ...]
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_golden import CASES, make_inputs  # noqa: E402
from oracle import run_reference as rr  # noqa: E402

OUT = os.path.join(HERE, "autocast_yardstick.json")
DEFAULT = ["small_padded", "L_D_coco80"]


def rel_rms(a, b):
    a, b = a.detach().float(), b.detach().float()
    fin = torch.isfinite(a) & torch.isfinite(b)
    a, b = a[fin], b[fin]
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def stage_table(S16, S32):
    """reference stage names -> the names the HIP stage taps use (tests/teacher_forced.py)"""
    t = {}
    for k, v in S32.items():
        if not torch.is_tensor(v) or k not in S16 or not v.is_floating_point() or tuple(v.shape) != tuple(S16[k].shape):
            continue
        name = k.replace("vit_block", "vit_blk")
        t[name] = rel_rms(S16[k], v)
    if "inter_states" in S32:
        for i in range(S32["inter_states"].shape[0]):
            t[f"dec{i}_out"] = rel_rms(S16["inter_states"][i], S32["inter_states"][i])
    if "inter_references" in S32:
        for i in range(S32["inter_references"].shape[0]):
            a, b = S16["inter_references"][i].float(), S32["inter_references"][i].float()
            t[f"dec{i}_ref_abs"] = float((a - b).pow(2).mean().sqrt())
    for k in ("pred_boxes", "init_reference"):
        if k in S32:
            t[k + "_abs"] = float((S16[k].float() - S32[k].float()).pow(2).mean().sqrt())
    return t


def main():
    torch.set_num_threads(8)
    cases = sys.argv[1:] or DEFAULT
    table = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for case in cases:
        cfg, wseed, image, text = make_inputs(case)
        t0 = time.time()
        S32, _, _, _ = rr.run_reference(cfg, wseed, image, text)
        t1 = time.time()
        S16, _, _, _ = rr.run_reference(cfg, wseed, image, text, autocast=torch.bfloat16, forced_topk=S32["topk_proposals"])
        t2 = time.time()
        table[case] = {"dtype": "bfloat16 (torch.autocast cpu)", "torch": torch.__version__, "stages": stage_table(S16, S32)}
        print(case, f"fp32 {t1 - t0:.0f} s, autocast {t2 - t1:.0f} s:", {k: f"{v:.2e}" for k, v in table[case]["stages"].items()
                                                                          if k in ("p2", "memory", "enc_class", "pred_logits", "pred_boxes_abs", "dec5_out", "dec1_out")})
        with open(OUT, "w") as fh:
            json.dump(table, fh, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
