"""The encoder's deformable-attention sampler on inputs shaped like the real ones: queries = the pixels of the five pyramid
levels of a 1024^2 image in raster order (87 296), reference points = their own centres, offsets = the reference's ring
initialisation (1..4 px in 8 directions per head, multi_scale_deform_attn.py:195-207) + N(0, sigma^2) noise, stored as IEEE half
like the production path; value = random bf16.  Graph-replayed timing; run under rocprofv3 --pmc for the counter passes.

    python tools/gpu_msda_case.py [--sigma 0.5] [--reps 20] [--size 1024]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops  # noqa: E402


def encoder_case(size=1024, sigma=0.5, seed=1, dev="cuda"):
    shapes = [(size // s, size // s) for s in (4, 8, 16, 32, 64)]
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    starts = [sum(h * w for h, w in shapes[:i]) for i in range(L)]
    g = torch.Generator().manual_seed(seed)
    value = torch.randn(S, 256, generator=g).to(torch.bfloat16).to(dev)
    refs = []
    for (h, w) in shapes:
        ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs)[:, None, :].repeat(1, L, 1).contiguous().to(dev)              # [Q, L, 2]
    th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
    grid = torch.stack([th.cos(), th.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(8, 1, 1, 2).repeat(1, L, 4, 1)
    for i in range(4):
        grid[:, :, i, :] *= i + 1
    off = grid.reshape(1, -1) + sigma * torch.randn(S, 8 * L * 4 * 2, generator=g)
    logit = torch.randn(S, 8 * L * 4, generator=g)
    offw = torch.cat([off, logit], 1).to(torch.float16).contiguous().to(dev)
    return value, shapes, starts, offw, ref, S


def bench(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sigma", type=float, nargs="*", default=[0.5])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--eager", action="store_true", help="plain launches (for rocprofv3 passes)")
    ap.add_argument("--check", action="store_true", help="compare every variant with the default kernel's output")
    args = ap.parse_args()
    variants = [v for v in os.environ.get("APE_MSDA_VARIANTS", "bf16,half").split(",") if v]      # value storage: bf16 | IEEE half
    for sigma in args.sigma:
        value, shapes, starts, offw, ref, S = encoder_case(args.size, sigma)
        out = torch.empty(S, 256, dtype=torch.bfloat16, device="cuda")
        alg = S * 256 * 2 + S * 480 * 2 + S * 256 * 2           # value + half offsets|logits + output
        base = None
        for v in variants:
            val = value.to(torch.float16) if v == "half" else value
            fn = lambda val=val: ops.msda_fused(val, shapes, starts, offw, ref, out=out)   # noqa: E731
            if args.eager:
                for _ in range(args.reps):
                    fn()
                torch.cuda.synchronize()
                print(f"sigma {sigma} variant {v}: {args.reps} eager launches done", flush=True)
                continue
            us = bench(fn, args.reps)
            msg = f"encoder sampler, {S} queries, offsets ring + N(0,{sigma}^2) px, variant {v}: {us:.1f} us  ({alg / us / 1e6:.2f} TB/s algorithmic of {alg / 1e6:.1f} MB)"
            if args.check:
                fn()
                torch.cuda.synchronize()
                if base is None:
                    base = out.float().clone()
                else:
                    err = ((out.float() - base).abs().max() / base.abs().max()).item()
                    msg += f"  max |diff| vs the first variant / max: {err:.2e}"
            print(msg, flush=True)


if __name__ == "__main__":
    main()
