"""Where does a step's time go: graph replay alone, + paste, + pipelined D2H, synchronous call."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ape_amd.modeling.build import build_ape, init_synthetic
from ape_amd.runtime import GraphedForward
import ape_amd.ops as ops

model = init_synthetic(build_ape("L_D"), 0).cuda()
mv = model.model_vision
mv.set_compute_dtype(torch.bfloat16)
image = torch.randint(0, 256, (3, 1024, 1024), generator=torch.Generator().manual_seed(2)).float().cuda()
text = torch.randn(80, 1024, generator=torch.Generator().manual_seed(3)).cuda()
run = GraphedForward(mv)
run(image, text); run(image, text)
e = next(iter(run._graphs.values()))
N = 30

def timeit(fn, n=N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print("graph replay only          : %.2f ms" % timeit(lambda: e.graph.replay()))
s = e.slots[0]
def rp():
    e.graph.replay(); ops.paste_bits(e.masks128, e.boxes, 1024, 1024, out=s.d_masks)
print("replay + paste             : %.2f ms" % timeit(rp))
print("synchronous __call__       : %.2f ms" % timeit(lambda: run(image, text)))
pend = [None]
def piped():
    t = run.submit(image, text)
    if pend[0] is not None: run.result(pend[0])
    pend[0] = t
ms = timeit(piped); run.result(pend[0])
print("pipelined submit/result    : %.2f ms" % ms)
# D2H alone
cs = torch.cuda.Stream()
def d2h():
    with torch.cuda.stream(cs):
        s.h_masks.copy_(s.d_masks, non_blocking=True)
    cs.synchronize()
print("D2H of the masks alone     : %.2f ms  (%.1f GB/s)" % (timeit(d2h, 10), s.d_masks.numel() / timeit(d2h, 10) / 1e6))
