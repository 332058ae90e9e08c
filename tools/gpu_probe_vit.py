"""Graph-replayed timing of the ViT-block GEMMs under every tiling the launcher offers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops

def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

bf = torch.bfloat16
M = 4096
cases = [("proj res,o32", 1024, 1024, "res32"), ("w3 res,o32", 1024, 2752, "res32"), ("qk rope", 2048, 1024, "rope"), ("v T", 1024, 1024, "T"),
         ("w1w2 swiglu", 5504, 1024, "swiglu"), ("qkv plain", 3072, 1024, "")]
for name, N, K, kind in cases:
    a = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf); b = torch.randn(N, device="cuda")
    kw = {}
    if kind == "res32": kw = dict(residual=torch.randn(M, N, device="cuda"), out_dtype=torch.float32)
    if kind == "rope":
        cos = torch.randn(M, 64, device="cuda"); sin = torch.randn(M, 64, device="cuda")
        kw = dict(rope=(cos, sin, M, 64, N))
    if kind == "T": kw = dict(trans_out=True)
    if kind == "swiglu": kw = dict(act=ops.ACT_SWIGLU)
    res = {}
    for t64 in (0, 1, 2):
        for sk in (1, 2):
            if sk > 1 and kind in ("T", "swiglu"): continue
            if t64 == 2 and kind == "T": continue
            try:
                res[(t64, sk)] = bench(lambda: ops.gemm(a, w, b, tile64=t64, splitk=sk, **kw))
            except Exception as ex:
                res[(t64, sk)] = float("nan")
    res["auto"] = bench(lambda: ops.gemm(a, w, b, **kw))
    os.environ["APE_GEMM_RING"] = "1"
    res["ring128"] = bench(lambda: ops.gemm(a, w, b, tile64=0, splitk=1, **kw))
    os.environ["APE_GEMM_RING"] = "0"
    best = min(v for v in res.values() if v == v)
    print(f"{name:14s} N{N} K{K}: " + "  ".join(f"{k}={v:.1f}" for k, v in res.items()), f" best {2*M*N*K/best/1e6:.0f} TF", flush=True)
