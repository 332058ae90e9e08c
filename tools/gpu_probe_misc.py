"""Graph-replayed timings of the non-GEMM kernels of one APE-L_D image (and a few K = 256 GEMMs): where is the time that
is not MFMA?  One line per kernel with its algorithmic HBM bytes and the rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / reps * 1e3)
    return sorted(ts)[1]


def line(name, us, nbytes):
    print(f"{name:58s} {us:9.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e6:6.2f} TB/s", flush=True)


bf, dev = torch.bfloat16, "cuda"
T = 87296
img = torch.randint(0, 256, (3, 1024, 1024)).float().to(dev)
t2r = torch.randperm(4096).int().to(dev)
for odt in (bf,):
    line("patchify 1024^2 -> [4096,768]", bench(lambda: ops.patchify(img, t2r, 64, 64, (123.0, 116.0, 103.0), (58.0, 57.0, 57.0), out_dtype=odt)), 12.6e6 + 6.3e6)
x32 = torch.randn(4096, 1024, device=dev)
xb = torch.randn(4096, 1024, device=dev).to(bf)
w, b = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
line("layernorm f32 [4096,1024] -> bf16", bench(lambda: ops.layernorm(x32, w, b, 1e-6, out_dtype=bf)), 4096 * 1024 * 6)
x32b = torch.randn(16384, 1024, device=dev)
line("layernorm f32 [16384,1024] -> bf16", bench(lambda: ops.layernorm(x32b, w, b, 1e-6, out_dtype=bf)), 16384 * 1024 * 6)
line("layernorm bf16 [4096,1024] -> bf16", bench(lambda: ops.layernorm(xb, w, b, 1e-6, out_dtype=bf)), 4096 * 1024 * 4)
h = torch.randn(4096, 2752, device=dev).to(bf)
line("row_stats bf16 [4096,2730]", bench(lambda: ops.row_stats(h[:, :2730], 1e-6)), 4096 * 2730 * 2)
e = torch.randn(T, 256, device=dev).to(bf)
pos = torch.randn(T, 256, device=dev).to(bf)
w2, b2 = torch.ones(256, device=dev), torch.zeros(256, device=dev)
line("layernorm bf16 [87296,256]", bench(lambda: ops.layernorm(e, w2, b2, 1e-5, out_dtype=bf)), T * 256 * 4)
line("layernorm bf16 [87296,256] + add -> 2 outputs", bench(lambda: ops.layernorm(e, w2, b2, 1e-5, out_dtype=bf, add=pos)), T * 256 * 8)
u = torch.randn(8, 256, device=dev).to(bf)
line("gemm [87296,8,256] f32 out (VL scores)", bench(lambda: ops.gemm(e, u, None, out_dtype=torch.float32)), T * 256 * 2 + T * 32)
S = torch.randn(T, 8, device=dev)
line("vl_pool [87296,8] x [87296,256]", bench(lambda: ops.vl_pool(S, e)), T * 256 * 2 + T * 32)
p2 = torch.randn(65536, 256, device=dev).to(bf)
line("im2col3x3 [65536,256] -> [65536,2304]", bench(lambda: ops.im2col3x3(p2, None, 256, 256)), 65536 * 256 * 2 * 10)
gw, gb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
line("groupnorm [65536,256]", bench(lambda: ops.groupnorm(p2, gw, gb, 32, 1e-5)), 65536 * 256 * 6)
ml = torch.randn(100, 65536, device=dev)
line("mask_upsample_bits 100 x 256^2 -> 1024^2 bits", bench(lambda: ops.mask_upsample_bits(ml, 256, 256, 1024), 5), 100 * 65536 * 4 + 100 * 131072)
bits = ops.mask_upsample_bits(ml, 256, 256, 1024)
boxes = (torch.rand(100, 4, device=dev) * 400)
boxes[:, 2:] += boxes[:, :2] + 50
m128 = ops.roi_align_bits(bits, boxes.contiguous(), 128)
line("roi_align_bits 100 boxes -> 128^2", bench(lambda: ops.roi_align_bits(bits, boxes.contiguous(), 128), 5), 100 * 16384)
outm = torch.empty(100, 1024, 1024, dtype=torch.uint8, device=dev)
line("paste_bits 100 x 1024^2", bench(lambda: ops.paste_bits(m128, boxes.contiguous(), 1024, 1024, out=outm), 5), 100 * 1048576)
for (M, N, K, kw, name) in [(T, 256, 256, {}, "value_proj"), (T, 480, 256, dict(out_dtype=torch.float32), "offsets+logits f32"),
                            (T, 480, 256, {}, "offsets+logits bf16"), (T, 2048, 256, dict(act=ops.ACT_RELU), "FFN1"),
                            (T, 1536, 256, {}, "decoder value_proj x6")]:
    a = torch.randn(M, K, device=dev).to(bf)
    ww = (torch.randn(N, K, device=dev) / 16).to(bf)
    bb = torch.randn(N, device=dev)
    osz = 4 if kw.get("out_dtype") == torch.float32 else 2
    us = bench(lambda: ops.gemm(a, ww, bb, **kw))
    print(f"gemm {name:24s} {M}x{N}x{K}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s  {(M * K * 2 + M * N * osz) / us / 1e6:5.2f} TB/s", flush=True)
