"""Round 6: what the in-launch row statistics (ApeGemmArgs.rowstat_cols, the RSTAT flavour of the 256 x 128 tile kernel) buy on the ViT's two
folded LayerNorms, and what the K = N = 256 LayerNorm-epilogue kernel costs without its scratch.

  down projection  8192 x 1024 x 2752 + fp32 residual:   row_stats + gemm(rownorm)        vs   gemm(rowstats)
  out projection   8192 x 1024 x 1024 + fp32 residual:   layernorm + gemm                 vs   gemm(rowstats) on the folded weight
  kres_ln          87296 x 256 x 256 + residual + LayerNorm epilogue

Each variant: `reps` back-to-back launches inside one event pair with cold-ish operands rotated through 4 buffer sets, rounds
interleaved, median reported; results compared."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops  # noqa: E402

DEV = torch.device("cuda")


def timeit(fns, reps=20, rounds=7):
    t = {k: [] for k in fns}
    for _ in range(rounds):
        for k, fn in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(reps):
                fn(i)
            e.record()
            e.synchronize()
            t[k].append(s.elapsed_time(e) * 1e3 / reps)
    return {k: statistics.median(v) for k, v in t.items()}


def main():
    for dt in (torch.bfloat16, torch.float16):
        g = torch.Generator().manual_seed(0)
        NB = 4
        # ---- down projection
        M, C, Cp, N = 8192, 2730, 2752, 1024
        hs = []
        for _ in range(NB):
            h = torch.zeros(M, Cp)
            h[:, :C] = torch.randn(M, C, generator=g) * 1.5 + 0.2
            hs.append(h.to(dt).to(DEV))
        wf = torch.zeros(N, Cp)
        wf[:, :C] = torch.randn(N, C, generator=g) / C ** 0.5
        wf = wf.to(dt).to(DEV)
        c1, c2 = wf.float().sum(1).contiguous(), torch.randn(N, generator=g).to(DEV)
        res = [torch.randn(M, N, generator=g).to(DEV) for _ in range(NB)]
        out = torch.empty((M, N), dtype=torch.float32, device=DEV)

        def two(i):
            st = ops.row_stats(hs[i % NB][:, :C], 1e-6)
            ops.gemm(hs[i % NB], wf, c2, residual=res[i % NB], rownorm=(st[0], st[1], c1), out=out)

        def one(i):
            ops.gemm(hs[i % NB], wf, c2, residual=res[i % NB], rowstats=(C, 1e-6, c1), out=out)

        def plain(i):
            ops.gemm(hs[i % NB], wf, c2, residual=res[i % NB], out=out)

        two(0); a = out.clone(); one(0); b = out.clone()
        err = ((a - b).norm() / a.norm()).item()
        r = timeit({"row_stats + gemm(rownorm)": two, "gemm(rowstats)": one, "gemm without the LayerNorm terms": plain})
        print(f"[{str(dt)[6:]}] down projection {M}x{N}x{Cp} + fp32 residual: " + ", ".join(f"{k} {v:.1f} us" for k, v in r.items()) + f"; difference {err:.2e}")
        # ---- out projection
        C = N = 1024
        os_ = [(torch.randn(M, C, generator=g) * 0.7 + 0.1).to(dt).to(DEV) for _ in range(NB)]
        gam, bet = (1.0 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
        w = (torch.randn(N, C, generator=g) / C ** 0.5).to(DEV)
        bias = torch.randn(N, generator=g).to(DEV)
        wp, wpf = w.to(dt), (w * gam[None, :]).to(dt)
        c1, c2 = wpf.float().sum(1).contiguous(), (w @ bet + bias).contiguous()

        def two(i):
            on = ops.layernorm(os_[i % NB], gam, bet, 1e-6, out_dtype=dt)
            ops.gemm(on, wp, bias, residual=res[i % NB], out=out)

        def one(i):
            ops.gemm(os_[i % NB], wpf, c2, residual=res[i % NB], rowstats=(C, 1e-6, c1), out=out)

        def plain(i):
            ops.gemm(os_[i % NB], wp, bias, residual=res[i % NB], out=out)

        two(0); a = out.clone(); one(0); b = out.clone()
        err = ((a - b).norm() / (a - res[0]).norm()).item()
        r = timeit({"layernorm + gemm": two, "gemm(rowstats) on the folded weight": one, "gemm alone": plain})
        print(f"[{str(dt)[6:]}] out projection {M}x{N}x{C} + fp32 residual: " + ", ".join(f"{k} {v:.1f} us" for k, v in r.items()) + f"; difference {err:.2e} (one 16-bit store fewer)")
        # ---- K = N = 256 with the LayerNorm epilogue
        M, C = 87296, 256
        xs = [torch.randn(M, C, generator=g).to(dt).to(DEV) for _ in range(NB)]
        rs = [torch.randn(M, C, generator=g).to(dt).to(DEV) for _ in range(NB)]
        w = (torch.randn(C, C, generator=g) / 16).to(dt).to(DEV)
        bias, gam, bet = torch.randn(C, generator=g).to(DEV), torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        o2 = torch.empty((M, C), dtype=dt, device=DEV)
        r = timeit({"kres_ln": lambda i: ops.gemm(xs[i % NB], w, bias, residual=rs[i % NB], norm=(gam, bet, 1e-5), out=o2)})
        print(f"[{str(dt)[6:]}] kres_ln {M}x{C}x{C} + residual + LayerNorm epilogue: {r['kres_ln']:.1f} us")


if __name__ == "__main__":
    main()
