"""One GEMM shape launched eagerly N times (for rocprofv3 PMC passes): python gpu_one_gemm.py M N K [kind] [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else ""
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
bf = torch.bfloat16
a = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf); b = torch.randn(N, device="cuda")
kw = {}
if kind == "rope":
    kw = dict(rope=(torch.randn(M, 64, device="cuda"), torch.randn(M, 64, device="cuda"), M, 64, N))
if kind == "swiglu": kw = dict(act=ops.ACT_SWIGLU)
if kind == "relu": kw = dict(act=ops.ACT_RELU)
for _ in range(reps):
    ops.gemm(a, w, b, **kw)
torch.cuda.synchronize()
