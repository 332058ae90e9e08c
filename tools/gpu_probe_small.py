"""Graph-replayed micro-benchmark of small / skinny GEMM shapes (host launch overhead excluded)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops

def bench(fn, reps=50):
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
for (M, N, K) in [(900, 256, 256), (900, 512, 256), (900, 2048, 256), (900, 256, 2048), (900, 480, 256), (900, 4, 256), (100, 65536, 256),
                  (4096, 1024, 1024), (4096, 2048, 1024), (4096, 5504, 1024), (4096, 1024, 2752), (87296, 256, 256), (87296, 2048, 256), (87296, 256, 2048)]:
    a = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=bf)
    res = {}
    for t64 in (0, 1):
        for sk in (1, 2, 4, 8):
            if sk > 1 and (K // 32 < sk * 2 or N % 4): continue
            res[(t64, sk)] = bench(lambda: ops.gemm(a, w, b, out=out, splitk=sk, tile64=t64))
    res["auto"] = bench(lambda: ops.gemm(a, w, b, out=out))
    print(f"M{M} N{N} K{K}: " + "  ".join(f"{k}={v:.1f}" for k, v in res.items()), f" ({2*M*N*K/min(res.values())/1e6:.0f} TF best)", flush=True)
# launch floor: an (almost) empty kernel
x = torch.zeros(64, 64, device="cuda").to(bf); wz = torch.zeros(8, 64, device="cuda").to(bf); o = torch.empty(64, 8, device="cuda", dtype=bf)
print("floor M64 N8 K64:", bench(lambda: ops.gemm(x, wz, None, out=o, splitk=1)))
y = torch.empty(64, 64, device="cuda")
print("floor torch add:", bench(lambda: torch.add(y, 1.0, out=y)))
