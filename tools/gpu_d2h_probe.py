"""How does the 105 MB mask transfer (device -> pinned host) travel: as a shader blit (__amd_rocclr_copyBuffer occupies CUs) or on an SDMA
engine?    Runs the copy through torch's copy_ and through hipMemcpyAsync / hipMemcpyDtoHAsync directly, alone
and next to a CU-saturating GEMM loop; run it under `rocprofv3 --kernel-trace --stats` per environment variant (tools/gpu_call.sh d2h)
to see whether a copy kernel shows up.  Prints one line per variant."""
import ctypes
import os
import time

import torch

hip = ctypes.CDLL("libamdhip64.so")
N = 100 * 1024 * 1024
x = torch.zeros(N, dtype=torch.uint8, device="cuda")
h = torch.empty(N, dtype=torch.uint8, pin_memory=True)
y = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
cs = torch.cuda.Stream()
torch.cuda.synchronize()


def copy_torch():
    with torch.cuda.stream(cs):
        h.copy_(x, non_blocking=True)


def copy_hip():
    rc = hip.hipMemcpyAsync(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_size_t(N), ctypes.c_int(2), ctypes.c_void_p(cs.cuda_stream))
    assert rc == 0, rc


def copy_dtoh():
    rc = hip.hipMemcpyDtoHAsync(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_size_t(N), ctypes.c_void_p(cs.cuda_stream))
    assert rc == 0, rc


def t_copy(fn, n=5):
    fn(); cs.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    cs.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def t_mm(n=40, fn=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fn is not None:
        fn()
    for _ in range(n):
        (y @ y)
    torch.cuda.current_stream().synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    return dt


def copy_sdma():
    """the library's DMA-engine entry (csrc/hostcopy.cpp): blocking, so the overlapped measurement runs it on a helper thread"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from ape_amd import _lib
    lib = _lib.load()
    rc = lib.ape_hip_sdma_d2h(h.data_ptr(), x.data_ptr(), N)
    assert rc == 0, lib.ape_hip_last_error()


envs = {k: os.environ.get(k) for k in ("HSA_ENABLE_SDMA", "GPU_FORCE_BLIT_COPY_SIZE", "HSA_FORCE_SDMA_SIZE", "HSA_ENABLE_SDMA_COPY_SIZE_OVERRIDE")}
print("env", envs)
t_mm(5)
base = min(t_mm() for _ in range(3))
import threading
x.fill_(7)
torch.cuda.synchronize()
h.zero_()
copy_sdma()
print("sdma copy correct:", bool((h == 7).all()))
for name, fn in (("torch.copy_", copy_torch), ("hipMemcpyAsync", copy_hip), ("hipMemcpyDtoHAsync", copy_dtoh), ("ape_hip_sdma_d2h", copy_sdma)):
    alone = t_copy(fn)
    if fn is copy_sdma:
        def threaded():
            th = threading.Thread(target=lambda: [fn(), fn(), fn()])
            th.start()
            threaded.th = th
        both = []
        for _ in range(3):
            both.append(t_mm(fn=threaded))
            threaded.th.join()
        both = min(both)
    else:
        both = min(t_mm(fn=lambda: [fn(), fn(), fn()]) for _ in range(3))
    print(f"{name:20s} copy alone {alone:6.2f} ms ({N / alone / 1e6:5.1f} GB/s)   40 GEMMs alone {base:7.2f} ms, with 3 copies in flight {both:7.2f} ms "
          f"(+{100 * (both / base - 1):.1f} %)")
