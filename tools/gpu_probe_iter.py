"""Per-K-iteration cost of the 128x128 GEMM kernels: time vs K at fixed grid (slope = cost of one BK=64 step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops

def bench(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

bf = torch.bfloat16
for (M, N) in [(1024, 1024), (2048, 2048), (4096, 2048), (4096, 4096), (8192, 8192)]:
    row = []
    for K in (512, 1024, 2048, 4096, 8192):
        a = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
        out = torch.empty(M, N, device="cuda", dtype=bf)
        os.environ["APE_GEMM_RING"] = "0"
        t_v2 = bench(lambda: ops.gemm(a, w, None, out=out, tile64=0, splitk=1))
        os.environ["APE_GEMM_RING"] = "1"
        t_ring = bench(lambda: ops.gemm(a, w, None, out=out, tile64=0, splitk=1))
        os.environ["APE_GEMM_RING"] = "0"
        row.append((K, t_v2, t_ring))
    blocks = (M // 128) * (N // 128)
    s_v2 = (row[-1][1] - row[1][1]) / ((row[-1][0] - row[1][0]) / 64)
    s_rg = (row[-1][2] - row[1][2]) / ((row[-1][0] - row[1][0]) / 64)
    print(f"M{M} N{N} ({blocks} tiles): " + "  ".join(f"K{k}: v2 {a:.1f} ring {b:.1f}" for k, a, b in row) +
          f"  | us per 64-k step: v2 {s_v2:.3f} ring {s_rg:.3f};  big-K TF: v2 {2*M*N*row[-1][0]/row[-1][1]/1e6:.0f} ring {2*M*N*row[-1][0]/row[-1][2]/1e6:.0f}", flush=True)
