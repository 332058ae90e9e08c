import os, sys, time, torch
x = torch.zeros(100, 1024, 1024, dtype=torch.uint8, device="cuda")
h = torch.empty(100, 1024, 1024, dtype=torch.uint8, pin_memory=True)
y = torch.randn(4096, 4096, device="cuda")
cs = torch.cuda.Stream()
torch.cuda.synchronize()
def t_copy():
    t0 = time.perf_counter()
    with torch.cuda.stream(cs):
        h.copy_(x, non_blocking=True)
    cs.synchronize()
    return (time.perf_counter() - t0) * 1e3
def t_mm(n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): (y @ y)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / n
t_copy(); t_mm()
print("env HSA_ENABLE_SDMA=%s" % os.environ.get("HSA_ENABLE_SDMA"), "copy alone %.2f ms" % t_copy(), "matmul alone %.3f ms" % t_mm())
# overlapped
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(cs):
    h.copy_(x, non_blocking=True)
for _ in range(20): (y @ y)
torch.cuda.synchronize()
print("copy + 20 matmuls overlapped: %.2f ms (sum of parts %.2f)" % ((time.perf_counter() - t0) * 1e3, t_copy() + 20 * t_mm()))
