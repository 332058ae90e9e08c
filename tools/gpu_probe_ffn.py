"""EXPERIMENTAL fused FFN (csrc/ffn_fused.hip) vs the two-GEMM form on the encoder's shape: first a correctness check on a small
and on the full problem (run this under a short `timeout`: the kernel was written without GPU time to validate it), then
graph-replayed timings and, with APE_FFN_FUSED=1, the whole bench."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ape_amd.ops as ops  # noqa: E402
import ref_ops  # noqa: E402


def bench(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    for M, HID in [(256, 128), (4096, 2048), (87296, 2048)]:
        x = torch.randn(M, 256, generator=g).to(bf).cuda()
        w1, b1 = (torch.randn(HID, 256, generator=g) / 16).to(bf).cuda(), torch.randn(HID, generator=g).cuda()
        w2, b2 = (torch.randn(256, HID, generator=g) * HID ** -0.5).to(bf).cuda(), torch.randn(256, generator=g).cuda()
        got = ops.ffn_fused(x, w1, b1, w2, b2, residual=x)
        torch.cuda.synchronize()
        ref = ref_ops.ffn_fused(x, w1, b1, w2, b2, residual=x)
        err = ((got.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
        print(f"ffn_fused M{M} HID{HID}: relerr vs the two-GEMM definition {err:.3e}", flush=True)
        if err > 6e-3:
            print("MISMATCH -- stop here")
            return
    from ape_amd.packing import permute_ffn_w2
    w2p = permute_ffn_w2(w2)
    t_f = bench(lambda: ops.ffn_fused(x, w1, b1, w2, b2, residual=x))
    t_p = bench(lambda: ops.ffn_fused(x, w1, b1, w2p, b2, residual=x, w2_permuted=True))
    for rt in ("2", "3"):
        os.environ["APE_FFN_RT"] = rt
        y = ops.ffn_fused(x, w1, b1, w2p, b2, residual=x, w2_permuted=True)
        err = ((y.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
        t_rt = bench(lambda: ops.ffn_fused(x, w1, b1, w2p, b2, residual=x, w2_permuted=True))
        print(f"   pre-permuted W2, {int(rt) * 64}-row workgroups (RT = {rt}): {t_rt:.1f} us ({2.0 * M * 256 * HID * 2 / t_rt / 1e6:.0f} TF/s), relerr {err:.2e}")
    os.environ.pop("APE_FFN_RT")
    t_2 = bench(lambda: ops.gemm(ops.gemm(x, w1, b1, act=ops.ACT_RELU), w2, b2, residual=x))
    fl = 2.0 * M * 256 * HID * 2
    print(f"87296 x 256 -> 2048 -> 256: fused, row-major W2 {t_f:.1f} us ({fl / t_f / 1e6:.0f} TF/s)   fused, pre-permuted W2 {t_p:.1f} us "
          f"({fl / t_p / 1e6:.0f} TF/s)   two GEMMs {t_2:.1f} us")


if __name__ == "__main__":
    main()
