"""Per-shape GEMM table of one APE-L_D image: every ops.gemm call timed with HIP events (eager, after warm-up)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops  # noqa: E402
from ape_amd.modeling.build import build_ape, init_synthetic  # noqa: E402


def main():
    model = init_synthetic(build_ape("L_D"), 0).cuda()
    mv = model.model_vision
    mv.set_compute_dtype(torch.bfloat16)
    image = torch.randint(0, 256, (3, 1024, 1024), generator=torch.Generator().manual_seed(2)).float().cuda()
    text = torch.randn(80, 1024, generator=torch.Generator().manual_seed(3)).cuda()
    for _ in range(2):
        mv.forward_single(image, text)
    torch.cuda.synchronize()
    real = ops.gemm
    rec = []

    def timed(a, w, bias=None, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = real(a, w, bias, **kw)
        e1.record()
        tag = []
        if kw.get("rope") is not None: tag.append("rope")
        if kw.get("trans_out"): tag.append("T")
        if kw.get("act"): tag.append(f"act{kw['act']}")
        if kw.get("residual") is not None: tag.append("res")
        if kw.get("rowmask") is not None: tag.append("mask")
        if a.dtype != torch.bfloat16: tag.append("f32")
        odt = out.dtype if torch.is_tensor(out) else None
        if odt == torch.float32: tag.append("o32")
        rec.append(((a.shape[0], w.shape[0], a.shape[1], ",".join(tag)), e0, e1))
        return out

    ops.gemm = timed
    reps = 3
    for _ in range(reps):
        mv.forward_single(image, text)
    torch.cuda.synchronize()
    ops.gemm = real
    agg = collections.defaultdict(list)
    for key, e0, e1 in rec:
        agg[key].append(e0.elapsed_time(e1) * 1e3)
    tot = 0.0
    print(f"{'M':>7} {'N':>6} {'K':>6} {'tags':18s} {'n/img':>5} {'us':>8} {'TF/s':>7} {'ms/img':>7}")
    for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        M, N, K, tag = key
        us = sum(v) / len(v)
        n = len(v) / reps
        tot += us * n / 1e3
        tiling = ops._auto_tiling(M, N, K, torch.bfloat16, "T" in tag, 0) if hasattr(ops, "_auto_tiling") else ""
        print(f"{M:7d} {N:6d} {K:6d} {tag:18s} {n:5.0f} {us:8.1f} {2.0 * M * N * K / us / 1e6:7.1f} {us * n / 1e3:7.3f}  {tiling}")
    print("total (event-timed, includes launch gaps) ms/img:", round(tot, 2))


if __name__ == "__main__":
    main()
