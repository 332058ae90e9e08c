"""Micro-benchmarks of the individual HIP kernels at the APE-L_D shapes (run on the GPU box).

usage: python tools/gpu_probe.py [--out gpurun_out/probe.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3  # seconds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/probe.json")
    args = ap.parse_args()
    dev = "cuda"
    res = {"device": torch.cuda.get_device_name(0)}
    bf = torch.bfloat16

    gemm_shapes = [
        ("vit_qk", 4096, 2048, 1024), ("vit_v", 4096, 1024, 1024), ("vit_proj", 4096, 1024, 1024),
        ("vit_w12", 4096, 5504, 1024), ("vit_w3", 4096, 1024, 2752), ("enc_ffn1", 87296, 2048, 256),
        ("enc_ffn2", 87296, 256, 2048), ("enc_val", 87296, 256, 256), ("enc_offw", 87296, 480, 256),
        ("dec_val6", 87296, 1536, 256), ("mask", 900, 65536, 256), ("big", 8192, 8192, 8192),
    ]
    for name, M, N, K in gemm_shapes:
        a = torch.randn(M, K, device=dev).to(bf)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=bf)
        t = timeit(lambda: ops.gemm(a, w, bias, out=out))
        res[f"gemm_{name}"] = {"M": M, "N": N, "K": K, "us": t * 1e6, "TFLOPs": 2.0 * M * N * K / t / 1e12}
        print(name, res[f"gemm_{name}"], flush=True)
        del a, w, out

    # torch (hipBLASLt) reference rate for the same shapes, for orientation only
    for name, M, N, K in gemm_shapes[:6]:
        a = torch.randn(M, K, device=dev).to(bf)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
        t = timeit(lambda: torch.matmul(a, w.t()))
        res[f"torchmm_{name}"] = {"us": t * 1e6, "TFLOPs": 2.0 * M * N * K / t / 1e12}
        print("torch", name, res[f"torchmm_{name}"], flush=True)

    for name, B, N, H, HD in [("win", 4, 1024, 16, 64), ("glb", 1, 4096, 16, 64), ("dec", 1, 900, 8, 32)]:
        E = H * HD
        q = torch.randn(B * N, E, device=dev).to(bf)
        k = torch.randn(B * N, E, device=dev).to(bf)
        npad = (B * N + 63) // 64 * 64
        vt = torch.zeros(E, npad, device=dev, dtype=bf)
        vt[:, : B * N] = torch.randn(E, B * N, device=dev).to(bf)
        o = torch.empty(B * N, E, device=dev, dtype=bf)
        t = timeit(lambda: ops.attention(q, k, vt, batch=B, n=N, heads=H, head_dim=HD, scale=HD ** -0.5, out=o))
        fl = 4.0 * B * H * N * N * HD
        res[f"attn_{name}"] = {"us": t * 1e6, "TFLOPs": fl / t / 1e12}
        print("attn", name, res[f"attn_{name}"], flush=True)

    shapes = [(256, 256), (128, 128), (64, 64), (32, 32), (16, 16)]
    S = sum(h * w for h, w in shapes)
    starts = [0, 65536, 81920, 86016, 87040]
    value = torch.randn(S, 256, device=dev).to(bf)
    # encoder-like queries: reference point = own cell centre, small offsets
    refs = []
    for h, w in shapes:
        ys, xs = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
        refs.append(torch.stack([(xs.flatten() + 0.5) / w, (ys.flatten() + 0.5) / h], -1))
    ref = torch.cat(refs)[:, None, :].repeat(1, 5, 1).contiguous().float()
    offw = torch.cat([torch.randn(S, 320, device=dev) * 2.0, torch.randn(S, 160, device=dev)], 1).contiguous()
    out = torch.empty(S, 256, device=dev, dtype=bf)
    t = timeit(lambda: ops.msda_fused(value, shapes, starts, offw, ref, out=out))
    alg = 173.2e6
    res["msda_enc"] = {"us": t * 1e6, "GBps_alg173MB": alg / t / 1e9}
    print("msda_enc", res["msda_enc"], flush=True)
    ref4 = torch.cat([torch.rand(900, 5, 2, device=dev), torch.rand(900, 5, 2, device=dev) * 0.3], -1).contiguous()
    offw4 = torch.cat([torch.randn(900, 320, device=dev) * 2.0, torch.randn(900, 160, device=dev)], 1).contiguous()
    out4 = torch.empty(900, 256, device=dev, dtype=bf)
    t = timeit(lambda: ops.msda_fused(value, shapes, starts, offw4, ref4, out=out4))
    res["msda_dec"] = {"us": t * 1e6}
    print("msda_dec", res["msda_dec"], flush=True)

    x = torch.randn(4096, 1024, device=dev)
    w, b = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
    y = torch.empty(4096, 1024, device=dev, dtype=bf)
    t = timeit(lambda: ops.layernorm(x, w, b, 1e-6, out=y))
    res["ln_vit"] = {"us": t * 1e6, "GBps": (x.numel() * 4 + y.numel() * 2) / t / 1e9}
    print("ln_vit", res["ln_vit"], flush=True)
    x = torch.randn(87296, 256, device=dev).to(bf)
    w, b = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    y = torch.empty(87296, 256, device=dev, dtype=bf)
    t = timeit(lambda: ops.layernorm(x, w, b, 1e-5, out=y))
    res["ln_enc"] = {"us": t * 1e6, "GBps": (x.numel() * 2 + y.numel() * 2) / t / 1e9}
    print("ln_enc", res["ln_enc"], flush=True)
    x = torch.randn(65536, 256, device=dev).to(bf)
    t = timeit(lambda: ops.groupnorm(x, w, b, 32, 1e-5))
    res["gn_p2"] = {"us": t * 1e6}
    print("gn_p2", res["gn_p2"], flush=True)

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
