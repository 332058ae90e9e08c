"""Graph-replayed comparison of the K=256 register-resident-A GEMM (kres) with the tile kernels it replaces."""
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
K = 256
for (M, N, kw_name) in [(87296, 2048, "relu"), (87296, 1536, "mask"), (87296, 480, "o32"), (87296, 480, ""), (87296, 256, "res"), (87296, 256, "mask"),
                        (87296, 256, "relu"), (87296, 512, "relu"), (65536, 256, ""), (16384, 256, ""), (4096, 256, "")]:
    a = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
    b = torch.randn(N, device="cuda")
    kw = {}
    odt = bf
    if kw_name == "relu": kw["act"] = ops.ACT_RELU
    if kw_name == "res": kw["residual"] = torch.randn(M, N, device="cuda").to(bf)
    if kw_name == "mask": kw.update(rowmask=(torch.arange(M, device="cuda") % 9 == 0).to(torch.uint8), mask_mode=ops.MASK_ZERO_OUTPUT)
    if kw_name == "o32": odt = torch.float32
    out = torch.empty(M, N, device="cuda", dtype=odt)
    res = {}
    os.environ["APE_GEMM_NOKRES"] = "1"
    res["auto(no kres)"] = bench(lambda: ops.gemm(a, w, b, out=out, **kw))
    ref = out.clone()
    res["ring128"] = bench(lambda: ops.gemm(a, w, b, out=out, tile64=0, splitk=1, **kw))
    os.environ["APE_GEMM_NOKRES"] = "0"
    out.zero_()
    res["kres"] = bench(lambda: ops.gemm(a, w, b, out=out, **kw))
    err = ((out.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
    byt = M * K * 2 + M * N * out.element_size() + (M * N * 2 if kw_name == "res" else 0)
    print(f"M{M} N{N} {kw_name:5s}: " + "  ".join(f"{k}={v:.1f}us" for k, v in res.items()),
          f" kres: {2*M*N*K/res['kres']/1e6:.0f} TF, {byt/res['kres']/1e6:.2f} TB/s, maxdiff vs tile kernel {err:.1e}", flush=True)
