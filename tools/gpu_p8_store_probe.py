"""EXPERIMENT: output store flavour of the eight-wave GEMM (APE_GEMM_STORE = plain | sc1 (write-through) | nt), on the ViT shapes of a
two-image step.  Graph-replayed, interleaved rounds, median."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ape_amd.ops as ops  # noqa: E402
from gpu_gemm_p8 import make_case, time_fn  # noqa: E402

SHAPES = [(8192, 2048, 1024, "rope"), (8192, 5504, 1024, "swiglu"), (8192, 1024, 2752, "res32"), (8192, 1024, 1024, "res32"),
          (65536, 256, 2304, "plain"), (16384, 5504, 1024, "swiglu")]


def main():
    dev = torch.device("cuda")
    print(f"{'M':>6} {'N':>5} {'K':>5} {'epi':7s} | " + " | ".join(f"{n:>14s}" for n in ("plain", "sc1", "nt")))
    for (M, N, K, kind) in SHAPES:
        a, w, bias, kw = make_case(M, N, K, kind, dev)
        out_n = N // 2 if kind == "swiglu" else N
        out = torch.empty((M, out_n), dtype=kw.get("out_dtype", torch.bfloat16), device=dev)
        kws = {k: v for k, v in kw.items() if k != "out_dtype"}
        reps = max(3, min(50, int(2e12 / (2.0 * M * N * K)) + 3))
        graphs, ref = {}, None
        for name in ("plain", "sc1", "nt"):
            os.environ["APE_GEMM_STORE"] = name
            ops.gemm(a, w, bias, out=out, **kws)
            torch.cuda.synchronize()
            cur = out.float().clone()
            if ref is None:
                ref = cur
            elif not torch.equal(cur, ref):
                print(f"   !! {name} output differs")
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(reps):
                    ops.gemm(a, w, bias, out=out, **kws)
            graphs[name] = g
        os.environ["APE_GEMM_STORE"] = "plain"
        times = {k: [] for k in graphs}
        for _ in range(5):
            for name, g in graphs.items():
                times[name].append(time_fn(g.replay, 1) / reps)
        cells = [f"{statistics.median(times[n]):7.1f}us {2.0 * M * N * K / statistics.median(times[n]) / 1e6:5.0f}T" for n in graphs]
        print(f"{M:6d} {N:5d} {K:5d} {kind:7s} | " + " | ".join(f"{c:>14s}" for c in cells), flush=True)


if __name__ == "__main__":
    main()
