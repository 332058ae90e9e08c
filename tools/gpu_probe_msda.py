"""Graph-replayed timing of the fused MSDA kernel on the encoder / decoder shapes (synthetic local offsets)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops

def bench(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

shapes = [(256, 256), (128, 128), (64, 64), (32, 32), (16, 16)]
S = sum(h * w for h, w in shapes)
ss = torch.tensor(shapes, dtype=torch.long)   # host copies: the C-ABI bakes the level table into the kernel arguments
lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
value = torch.randn(S, 256, device="cuda").to(torch.bfloat16)
for (Q, refdim, name) in [(S, 2, "encoder (87296 queries)"), (900, 4, "decoder (900 queries)")]:
    g = torch.Generator(device="cuda").manual_seed(1)
    off = torch.randn(Q, 8, 5, 4, 2, device="cuda", generator=g) * 2.0      # +-2 px offsets (local, like a trained model)
    logit = torch.randn(Q, 8 * 20, device="cuda", generator=g)
    offw = torch.cat([off.reshape(Q, -1), logit], 1).contiguous()
    ref = torch.rand(Q, 1, refdim, device="cuda", generator=g).repeat(1, 5, 1).contiguous()
    if refdim == 4: ref[..., 2:] = ref[..., 2:] * 0.3 + 0.05
    out = torch.empty(Q, 256, device="cuda", dtype=torch.bfloat16)
    us = bench(lambda: ops.msda_fused(value, ss, lsi, offw, ref, out=out))
    print(f"{name}: {us:.1f} us   ({(S * 256 * 2 + Q * 480 * 4 + Q * 256 * 2) / us / 1e6:.2f} TB/s algorithmic)", flush=True)
