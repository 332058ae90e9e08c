"""the two production attention launches of the 2-image ViT pass, run eagerly a few times (for counter passes)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops
bf = torch.bfloat16
for (B, N, H, HD) in [(2, 4096, 16, 64), (8, 1024, 16, 64)]:
    T = B * N
    q = torch.randn(T, H * HD, device="cuda").to(bf); k = torch.randn(T, H * HD, device="cuda").to(bf)
    vt = torch.randn(H * HD, T, device="cuda").to(bf)
    out = torch.empty(T, H * HD, device="cuda", dtype=bf)
    for _ in range(6):
        ops.attention(q, k, vt, batch=B, n=N, heads=H, head_dim=HD, scale=HD ** -0.5, out=out)
    torch.cuda.synchronize()
