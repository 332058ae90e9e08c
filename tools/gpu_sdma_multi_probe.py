"""Round 6: device -> pinned-host transfer of the 1536^2 / top-500 configuration's masks (2 x 1.18 GB per step) on the copy engines:
one blocking ape_hip_sdma_d2h per image (what the runtime did) against ape_hip_sdma_d2h_multi with 1 / 2 / 4 pieces per image and
engine placement on / off -- alone, and while the library's GEMM loop runs (the copy engines then compete with the kernels for HBM).
Every variant's bytes are verified."""
import ctypes
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops  # noqa: E402
from ape_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = torch.device("cuda")
NB = 500 * 1536 * 1536
src = [torch.randint(0, 255, (NB,), dtype=torch.uint8, device=DEV) for _ in range(2)]
dst = [torch.empty((NB,), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
for d in dst:
    d.zero_()
a = torch.randn(8192, 1024, device=DEV).to(torch.bfloat16)
w = torch.randn(5504, 1024, device=DEV).to(torch.bfloat16)
torch.cuda.synchronize()
print("copy engines free (device -> host):", lib.ape_hip_sdma_engines(dst[0].data_ptr(), src[0].data_ptr()))


def seq():
    for d, s in zip(dst, src):
        assert lib.ape_hip_sdma_d2h(d.data_ptr(), s.data_ptr(), NB) == 0


def multi(parts, engines):
    def f():
        os.environ["APE_SDMA_ENGINES"] = "1" if engines else "0"
        m = 2
        ds = (ctypes.c_void_p * m)(*[d.data_ptr() for d in dst])
        ss = (ctypes.c_void_p * m)(*[s.data_ptr() for s in src])
        sz = (ctypes.c_size_t * m)(NB, NB)
        assert lib.ape_hip_sdma_d2h_multi(m, ds, ss, sz, parts) == 0, lib.ape_hip_last_error()
    return f


def gemm_loop(stop, count):
    torch.cuda.set_device(0)
    while not stop.is_set():
        for _ in range(10):
            ops.gemm(a, w, None, act=ops.ACT_SWIGLU)
        torch.cuda.synchronize()
        count[0] += 10


variants = [("one blocking copy per image (round-6 runtime until now)", seq)] + [
    (f"multi, {p} piece(s) per image, engines {'placed' if e else 'left to the runtime'}", multi(p, e)) for p in (1, 2, 4) for e in (True, False)]
for name, fn in variants:
    for d in dst:
        d.zero_()
    fn()
    ok = all(torch.equal(d[:: 4097], s.cpu()[:: 4097]) for d, s in zip(dst, src)) and all(torch.equal(d[-4096:], s[-4096:].cpu()) for d, s in zip(dst, src))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    alone = min(ts)
    stop, count = threading.Event(), [0]
    th = threading.Thread(target=gemm_loop, args=(stop, count))
    th.start()
    time.sleep(0.3)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    stop.set()
    th.join()
    busy = min(ts)
    print(f"{name:78s} correct {ok}   alone {alone * 1e3:6.1f} ms ({2 * NB / alone / 1e9:5.1f} GB/s)   under a GEMM loop {busy * 1e3:6.1f} ms ({2 * NB / busy / 1e9:5.1f} GB/s)")
