"""A/B timing of the bf16 GEMM kernels on the shapes of one APE-L_D step (random-normal operands, not zeros).

For every shape: the tiling `ops._auto_tiling` picks today, the 256x256 and 256x128 eight-wave kernels (gemm_p8.hip) with
both barrier schedules, and torch.matmul (hipBLASLt) as a yardstick.  Each variant: `reps` back-to-back launches inside
one event pair, several rounds interleaved across variants, median reported.
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops  # noqa: E402

SHAPES = [
    # (M, N, K, kwargs-name) -- ViT block linears for 1 / 2 / 4 images, FPN / mask-head 3x3 convs, encoder FFN2
    (4096, 2048, 1024, "rope"), (4096, 5504, 1024, "swiglu"), (4096, 1024, 2752, "res32"), (4096, 1024, 1024, "res32"),
    (8192, 2048, 1024, "rope"), (8192, 5504, 1024, "swiglu"), (8192, 1024, 2752, "res32"),
    (16384, 2048, 1024, "rope"), (16384, 5504, 1024, "swiglu"), (16384, 1024, 2752, "res32"), (16384, 1024, 1024, "res32"),
    (16384, 1024, 1024, "trans"),
    (65536, 256, 2304, "plain"), (87296, 256, 2048, "res16"), (87296, 2048, 256, "relu"),
    (4096, 4096, 4096, "plain"), (8192, 8192, 8192, "plain"),
]


def make_case(M, N, K, kind, dev):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    kw = {}
    if kind == "rope":
        # pairs (2i, 2i + 1) share an angle like the ViT's tables: the packed (cos, sin) table rides along (ApeGemmArgs.rope_cs)
        c = torch.randn(4096, 32, generator=g).repeat_interleave(2, 1).contiguous().to(dev)
        sn = torch.randn(4096, 32, generator=g).repeat_interleave(2, 1).contiguous().to(dev)
        kw = dict(rope=(c, sn, 4096, 64, N, torch.stack([c[:, 0::2], sn[:, 0::2]], -1).contiguous()))
    elif kind == "swiglu":
        kw = dict(act=ops.ACT_SWIGLU)
    elif kind == "res32":
        kw = dict(residual=torch.randn(M, N, generator=g).to(dev), out_dtype=torch.float32)
    elif kind == "res16":
        kw = dict(residual=torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev))
    elif kind == "relu":
        kw = dict(act=ops.ACT_RELU)
    elif kind == "trans":
        kw = dict(trans_out=True)
    return a, w, bias, kw


def time_fn(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps   # us


def main():
    dev = torch.device("cuda")
    only = sys.argv[1:]
    print(f"{'M':>6} {'N':>5} {'K':>5} {'epi':7s} | " + " | ".join(f"{n:>14s}" for n in ("auto", "p8-256 s0", "p8-256 s1", "p8-128 s0", "p8-128 s1", "hipBLASLt")))
    for (M, N, K, kind) in SHAPES:
        if only and f"{M}x{N}x{K}" not in only:
            continue
        a, w, bias, kw = make_case(M, N, K, kind, dev)
        out_n = N // 2 if kind == "swiglu" else N
        odt = kw.get("out_dtype", torch.bfloat16)
        out = torch.empty((N, M) if kind == "trans" else (M, out_n), dtype=odt, device=dev)
        variants = {}
        variants["auto"] = (lambda: ops.gemm(a, w, bias, out=out, **{k: v for k, v in kw.items() if k != "out_dtype"}), None)
        for t, bn in ((3, 256), (4, 128)):
            for st in (0, 1):
                variants[f"p8-{bn} s{st}"] = (lambda t=t: ops.gemm(a, w, bias, out=out, tile64=t, **{k: v for k, v in kw.items() if k != "out_dtype"}), st)
        wt = w.t().contiguous()
        variants["hipBLASLt"] = (lambda: torch.matmul(a, wt), None)
        reps = max(3, min(50, int(2e12 / (2.0 * M * N * K)) + 3))
        times = {k: [] for k in variants}
        ref = None
        graphs = {}
        for name, (fn, st) in variants.items():            # `reps` launches per hipGraph: no host launch overhead in the timing
            if st is not None:
                os.environ["APE_GEMM_P8_STAGGER"] = str(st)
            fn()
            torch.cuda.synchronize()
            if name != "hipBLASLt":                        # agreement between our variants (same epilogue)
                cur = out.float().clone()
                if ref is None:
                    ref = cur
                else:
                    err = ((cur - ref).abs().max() / ref.abs().max()).item()
                    if err > 1e-2:
                        print(f"   !! {name} differs from auto by {err:.3e}")
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(reps):
                    fn()
            graphs[name] = g
        for rnd_ in range(5):
            for name in variants:
                times[name].append(time_fn(graphs[name].replay, 1) / reps)
        cells = []
        for name in variants:
            us = statistics.median(times[name])
            cells.append(f"{us:7.1f}us {2.0 * M * N * K / us / 1e6:5.0f}T")
        print(f"{M:6d} {N:5d} {K:5d} {kind:7s} | " + " | ".join(f"{c:>14s}" for c in cells), flush=True)


if __name__ == "__main__":
    main()
