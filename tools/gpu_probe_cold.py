"""Is a ViT GEMM slower inside the model than in a warm micro-benchmark because its operands are cold?  The SwiGLU up projection
(8192 x 5504 x 1024) and the q|k projection, timed per launch with HIP events: (hot) one operand set replayed, (cold W) 24 weight
sets cycled (270 MB > the Infinity Cache next to the activations), (cold all) 24 sets of A / W / C cycled with ~400 MB of unrelated
traffic between launches, (cold all + touch) the same with the NEXT launch's weights read once by a streaming kernel beforehand."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops  # noqa: E402
from gpu_gemm_p8 import make_case  # noqa: E402


def run(M, N, K, kind, nset, trash_mb, touch):
    dev = torch.device("cuda")
    sets = []
    for i in range(nset):
        a, w, bias, kw = make_case(M, N, K, kind, dev)
        if i:
            a, w = a.roll(i, 0).contiguous(), w.roll(i, 0).contiguous()
        out_n = N // 2 if kind == "swiglu" else N
        sets.append((a, w, bias, {k: v for k, v in kw.items() if k != "out_dtype"}, torch.empty((M, out_n), dtype=torch.bfloat16, device=dev)))
    trash = torch.empty(trash_mb << 20, dtype=torch.uint8, device=dev) if trash_mb else None
    times = []
    for it in range(4 * nset + 8):
        a, w, bias, kw, out = sets[it % nset]
        if trash is not None:
            trash.add_(1)                                   # unrelated traffic: reads + writes trash_mb each
        if touch:
            sets[it % nset][1].view(torch.int32).sum()      # streaming read of the weights about to be used
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.gemm(a, w, bias, out=out, tile64=3, **kw)
        e.record()
        e.synchronize()
        if it >= 8:
            times.append(s.elapsed_time(e) * 1e3)
    return statistics.median(times)


def main():
    for (M, N, K, kind) in [(8192, 5504, 1024, "swiglu"), (8192, 2048, 1024, "rope")]:
        res = {"hot": run(M, N, K, kind, 1, 0, False), "cold W": run(M, N, K, kind, 24, 0, False),
               "cold all": run(M, N, K, kind, 24, 200, False), "cold all + touch W": run(M, N, K, kind, 24, 200, True)}
        print(f"{M}x{N}x{K} {kind}: " + ", ".join(f"{k} {v:.1f} us" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
