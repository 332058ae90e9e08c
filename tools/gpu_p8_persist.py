"""Persistent launches of the 256-row tile kernel (csrc/gemm_p8.hip, tile loop): bit-identity against one workgroup per tile
(APE_P8_PERSIST=0) on every epilogue flavour that can go persistent, then the stand-alone duration of both (the library's launch
meter: each launch's own begin / end timestamps, a device synchronise between launches) and their back-to-back rate."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops  # noqa: E402
from ape_amd import _lib  # noqa: E402

CASES = [
    # M, N, K, kind, tile64, dtype
    (8192, 5504, 1024, "swiglu", 3, torch.bfloat16), (8192, 5504, 1024, "swiglu", 3, torch.float16), (8192, 5504, 1024, "swiglu_nobias", 3, torch.bfloat16),
    (16384, 5504, 1024, "swiglu", 3, torch.bfloat16), (8000, 5504, 1024, "swiglu", 3, torch.bfloat16), (8192, 5504, 1024, "swiglu_f32out", 3, torch.bfloat16),
    (32768, 1024, 1024, "res32", 3, torch.bfloat16), (16384, 2048, 1024, "rope", 3, torch.bfloat16), (87000, 2048, 256, "relu", 3, torch.bfloat16),
    (8192, 8192, 1024, "plain", 3, torch.bfloat16), (16384, 1024, 1024, "trans", 3, torch.bfloat16), (65536, 1024, 512, "plain", 4, torch.bfloat16),
    (8192, 2048, 1024, "rope", 3, torch.bfloat16), (8192, 4096, 64, "plain", 3, torch.bfloat16),
]


def make(M, N, K, kind, dt, dev):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dt).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dt).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    kw = {}
    if kind == "rope":
        c = torch.randn(4096, 32, generator=g).repeat_interleave(2, 1).contiguous().to(dev)
        sn = torch.randn(4096, 32, generator=g).repeat_interleave(2, 1).contiguous().to(dev)
        kw = dict(rope=(c, sn, 4096, 64, N, torch.stack([c[:, 0::2], sn[:, 0::2]], -1).contiguous()))
    elif kind.startswith("swiglu"):
        kw = dict(act=ops.ACT_SWIGLU)
        if kind == "swiglu_nobias":
            bias = None
        if kind == "swiglu_f32out":
            kw["out_dtype"] = torch.float32
    elif kind == "res32":
        kw = dict(residual=torch.randn(M, N, generator=g).to(dev), out_dtype=torch.float32)
    elif kind == "relu":
        kw = dict(act=ops.ACT_RELU)
    elif kind == "trans":
        kw = dict(trans_out=True)
    return a, w, bias, kw


def respf_table(dev, lib):
    """the fp32-residual prefetch of the 256 x 128 tile kernel (APE_P8_RESPF): stand-alone and back-to-back, cold operands"""
    import time
    print(f"\n{'M':>6} {'N':>5} {'K':>5} | residual prefetch: stand-alone us  off / on | back-to-back us  off / on")
    for (M, N, K) in [(8192, 1024, 1024), (8192, 1024, 2752 // 64 * 64), (16384, 1024, 1024)]:
        g = torch.Generator().manual_seed(M + K)
        R = 6
        As = [torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev) for _ in range(R)]
        Rs = [torch.randn(M, N, generator=g).to(dev) for _ in range(R)]
        Cs = [torch.empty(M, N, device=dev) for _ in range(R)]
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        st = [0]

        def run():
            i = st[0] = (st[0] + 1) % R
            return ops.gemm(As[i], w, bias, residual=Rs[i], out=Cs[i], tile64=4)
        alone, b2b = {}, {}
        ts = {"0": [], "1": []}
        for rnd in range(4):
            for mode in ("0", "1"):
                os.environ["APE_P8_RESPF"] = mode
                run(); torch.cuda.synchronize()
                lib.ape_hip_meter_begin()
                for _ in range(6):
                    run(); torch.cuda.synchronize()
                n = lib.ape_hip_meter_end()
                ms, nm = ctypes.c_float(), ctypes.c_char_p()
                for i in range(n):
                    lib.ape_hip_meter_read(i, ctypes.byref(nm), ctypes.byref(ms))
                    ts[mode].append(ms.value * 1e3)
        for mode in ("0", "1"):
            os.environ["APE_P8_RESPF"] = mode
            alone[mode] = sorted(ts[mode])[len(ts[mode]) // 2]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(); s.record()
            for _ in range(24):
                run()
            e.record(); e.synchronize()
            b2b[mode] = s.elapsed_time(e) * 1e3 / 24
        print(f"{M:6d} {N:5d} {K:5d} | {alone['0']:30.1f} / {alone['1']:.1f} | {b2b['0']:18.1f} / {b2b['1']:.1f}", flush=True)
    os.environ.pop("APE_P8_RESPF", None)


def main():
    dev = torch.device("cuda")
    lib = _lib.load()
    # a fresh box runs its first seconds at idle clocks (the first table of this probe read 2 x the bench's durations): 4 s of GEMMs first
    import time
    x = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    t0 = time.time()
    while time.time() - t0 < 4.0:
        for _ in range(20):
            ops.gemm(x, x, None, tile64=3)
        torch.cuda.synchronize()
    print(f"{'M':>6} {'N':>5} {'K':>5} {'kind':14s} {'dt':5s} | identical | stand-alone us: tile-per-wg  persistent | back-to-back us: tile-per-wg  persistent | TF/s persistent")
    for (M, N, K, kind, t64, dt) in CASES:
        a, w, bias, kw = make(M, N, K, kind, dt, dev)
        outs, alone, b2b = {}, {}, {}
        run = lambda: ops.gemm(a, w, bias, tile64=t64, **kw)
        ts = {"0": [], "1": []}
        for rnd in range(4):                         # modes interleaved: clock / thermal drift hits both alike
            for mode in ("0", "1"):
                os.environ["APE_P8_PERSIST"] = mode
                if rnd == 0:
                    outs[mode] = run().clone()
                    torch.cuda.synchronize()
                lib.ape_hip_meter_begin()
                for _ in range(6):
                    run()
                    torch.cuda.synchronize()
                n = lib.ape_hip_meter_end()
                ms, name = ctypes.c_float(), ctypes.c_char_p()
                for i in range(n):
                    lib.ape_hip_meter_read(i, ctypes.byref(name), ctypes.byref(ms))
                    ts[mode].append(ms.value * 1e3)
        for mode in ("0", "1"):
            os.environ["APE_P8_PERSIST"] = mode
            alone[mode] = sorted(ts[mode])[len(ts[mode]) // 2]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            run()
            s.record()
            for _ in range(reps):
                run()
            e.record()
            e.synchronize()
            b2b[mode] = s.elapsed_time(e) * 1e3 / reps
        same = torch.equal(outs["0"], outs["1"]) and bool(torch.isfinite(outs["1"].float()).all())
        print(f"{M:6d} {N:5d} {K:5d} {kind:14s} {str(dt)[6:]:5s} | {str(same):9s} | {alone['0']:12.1f} {alone['1']:11.1f} | {b2b['0']:12.1f} {b2b['1']:11.1f} | "
              f"{2.0 * M * N * K / alone['1'] / 1e6:7.0f}", flush=True)
        if not same:
            d = (outs["0"].float() - outs["1"].float()).abs()
            print(f"   MISMATCH: max abs diff {float(d.max()):.4g}, {int((d > 0).sum())} elements, first at {torch.nonzero(d > 0)[0].tolist()}")
    os.environ.pop("APE_P8_PERSIST", None)
    respf_table(dev, lib)


if __name__ == "__main__":
    main()
