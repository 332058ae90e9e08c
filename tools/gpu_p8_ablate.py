"""Where does a p8 launch spend its time?  Ablations of gemm_bf16_p8_kernel<256, true> (library built with -DAPE_P8_ABLATION:
`python tools/build_ablation.py`) on the step's shapes: 0 = the kernel, 1 = epilogue arithmetic + stores without the bias / RoPE
table / residual loads, 2 = full epilogue without stores, 3 = no epilogue, 4 = ONE K tile + full epilogue (everything but the
main loop).  hipGraph-replayed launches, interleaved rounds, median."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops  # noqa: E402
from gpu_gemm_p8 import make_case, time_fn  # noqa: E402

SHAPES = [(8192, 2048, 1024, "rope"), (8192, 5504, 1024, "swiglu"), (16384, 1024, 2752, "res32"), (16384, 1024, 1024, "res32"),
          (16384, 1024, 1024, "trans"), (65536, 256, 2304, "plain"), (16384, 2048, 1024, "rope")]
NAMES = ["kernel", "no epi loads", "no stores", "no epilogue", "1 K tile", "hipBLASLt"]


def main():
    dev = torch.device("cuda")
    print(f"{'M':>6} {'N':>5} {'K':>5} {'epi':7s} | " + " | ".join(f"{n:>12s}" for n in NAMES))
    for (M, N, K, kind) in SHAPES:
        a, w, bias, kw = make_case(M, N, K, kind, dev)
        out_n = N // 2 if kind == "swiglu" else N
        odt = kw.get("out_dtype", torch.bfloat16)
        out = torch.empty((N, M) if kind == "trans" else (M, out_n), dtype=odt, device=dev)
        kws = {k: v for k, v in kw.items() if k != "out_dtype"}
        wt = w.t().contiguous()
        reps = max(3, min(50, int(2e12 / (2.0 * M * N * K)) + 3))
        graphs, times = {}, {n: [] for n in NAMES}
        for abl, name in enumerate(NAMES):
            if name == "hipBLASLt":
                fn = lambda: torch.matmul(a, wt)      # noqa: E731
            else:
                os.environ["APE_P8_ABLATE"] = str(abl)
                fn = lambda: ops.gemm(a, w, bias, out=out, tile64=3, **kws)      # noqa: E731
            fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(reps):
                    fn()
            graphs[name] = g
        os.environ["APE_P8_ABLATE"] = "0"
        for _ in range(5):
            for name in NAMES:
                times[name].append(time_fn(graphs[name].replay, 1) / reps)
        print(f"{M:6d} {N:5d} {K:5d} {kind:7s} | " + " | ".join(f"{statistics.median(times[n]):10.1f}us" for n in NAMES), flush=True)


if __name__ == "__main__":
    main()
