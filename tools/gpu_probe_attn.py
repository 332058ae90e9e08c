"""Graph-replayed timing of the attention kernel on the model's shapes."""
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
for (B, N, H, HD, name) in [(4, 1024, 16, 64, "ViT windowed"), (1, 4096, 16, 64, "ViT global"), (8, 1024, 16, 64, "ViT windowed, 2 images"),
                            (2, 4096, 16, 64, "ViT global, 2 images"), (1, 900, 8, 32, "decoder self-attn"),
                            (9, 1024, 16, 64, "ViT windowed 1536^2"), (1, 9216, 16, 64, "ViT global 1536^2")]:
    T = B * N
    q = torch.randn(T, H * HD, device="cuda").to(bf); k = torch.randn(T, H * HD, device="cuda").to(bf)
    Tp = (T + 63) // 64 * 64
    vt = torch.zeros(H * HD, Tp, device="cuda", dtype=bf); vt[:, :T] = torch.randn(H * HD, T, device="cuda").to(bf)
    out = torch.empty(T, H * HD, device="cuda", dtype=bf)
    us = bench(lambda: ops.attention(q, k, vt, batch=B, n=N, heads=H, head_dim=HD, scale=HD ** -0.5, out=out))
    fl = 4.0 * B * H * N * N * HD
    print(f"{name:22s} B{B} N{N} H{H} hd{HD}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
