"""A/B timing of the K = 256 "kres" GEMM kernel on the encoder's shapes: rows per workgroup (APE_KRES_MI = 2 | 3) and column split
(APE_KRES_YSPLIT); each variant `reps` back-to-back launches inside one event pair, rounds interleaved, median reported; every
variant's output is checked against variant 0."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops  # noqa: E402

SHAPES = [(87296, 480, 256, torch.float16), (87296, 256, 256, torch.float16), (87296, 1536, 256, torch.float16), (87296, 256, 256, torch.bfloat16),
          (65536, 256, 256, torch.bfloat16), (21824, 480, 256, torch.float16)]
VARIANTS = [("mi2", {"APE_KRES_MI": "2"}), ("mi3", {"APE_KRES_MI": "3"}), ("mi2 y2", {"APE_KRES_MI": "2", "APE_KRES_YSPLIT": "2"}),
            ("mi2 y4", {"APE_KRES_MI": "2", "APE_KRES_YSPLIT": "4"}), ("mi3 y2", {"APE_KRES_MI": "3", "APE_KRES_YSPLIT": "2"})]


def run(fn, env):
    for k in ("APE_KRES_MI", "APE_KRES_YSPLIT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    return fn()


def main():
    dev = torch.device("cuda")
    print(f"{'M':>6} {'N':>5} {'out':>9} | " + " | ".join(f"{n:>9s}" for n, _ in VARIANTS) + " | default")
    for (M, N, K, odt) in SHAPES:
        g = torch.Generator().manual_seed(M + N)
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        out = torch.empty((M, N), dtype=odt, device=dev)
        fn = lambda: ops.gemm(a, w, bias, out=out)
        ref = None
        times = {n: [] for n, _ in VARIANTS}
        times["default"] = []
        for n, env in VARIANTS:
            run(fn, env)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
                want = (a.float() @ w.float().t() + bias).to(odt)
                err = ((ref.float() - want.float()).abs().max() / want.float().abs().max()).item()
                assert err < 1e-2, err
            else:
                assert torch.equal(out, ref), (n, (out.float() - ref.float()).abs().max().item())
        for _ in range(7):
            for n, env in VARIANTS + [("default", {})]:
                def timed():
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(20):
                        fn()
                    e.record()
                    e.synchronize()
                    return s.elapsed_time(e) * 1e3 / 20
                times[n].append(run(timed, env))
        print(f"{M:>6} {N:>5} {str(odt)[6:]:>9} | " + " | ".join(f"{statistics.median(times[n]):>9.1f}" for n, _ in VARIANTS) + f" | {statistics.median(times['default']):.1f}   (last kernel: {ops._lib.load().ape_hip_gemm_last_kernel().decode() if hasattr(ops._lib.load(), 'ape_hip_gemm_last_kernel') else ''})")


if __name__ == "__main__":
    main()
