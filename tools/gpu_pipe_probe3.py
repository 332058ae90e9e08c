"""Which piece of the per-image wrapper around the graph replay costs the extra ~0.8 ms?"""
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
s = e.slots[0]
N = 30
def timeit(fn, n=N):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("replay only                      : %.2f ms" % timeit(lambda: e.graph.replay()))
def a():
    e.image.copy_(image, non_blocking=True); e.graph.replay()
print("image.copy_ + replay             : %.2f ms" % timeit(a))
def a2():
    torch.add(image, 0.0, out=e.image); e.graph.replay()
print("image via add kernel + replay    : %.2f ms" % timeit(a2))
def b():
    e.image.copy_(image, non_blocking=True); e.graph.replay(); s.d_rec.copy_(e.rec, non_blocking=True)
    ops.paste_bits(e.masks128, e.boxes, 1024, 1024, out=s.d_masks)
print("+ rec copy + paste               : %.2f ms" % timeit(b))
cs = torch.cuda.Stream()
ev1, ev2 = torch.cuda.Event(), torch.cuda.Event()
def c():
    b()
    cur = torch.cuda.current_stream()
    ev1.record(cur)
    with torch.cuda.stream(cs):
        cs.wait_event(ev1)
        s.h_rec.copy_(s.d_rec, non_blocking=True)
        s.h_masks.copy_(s.d_masks, non_blocking=True)
        ev2.record(cs)
print("+ events + D2H on copy stream    : %.2f ms" % timeit(c))
def d():
    cur = torch.cuda.current_stream()
    cur.wait_event(ev2)
    c()
print("+ wait_event on slot reuse       : %.2f ms" % timeit(d))

for nb in (8, 32, 128):
    def c2():
        b()
        cur = torch.cuda.current_stream()
        ev1.record(cur)
        with torch.cuda.stream(cs):
            cs.wait_event(ev1)
            s.h_rec.copy_(s.d_rec, non_blocking=True)
            ops.copy_to_pinned(s.h_masks, s.d_masks, max_blocks=nb)
            ev2.record(cs)
    print("+ D2H by a %3d-block copy KERNEL   : %.2f ms" % (nb, timeit(c2)))
def d2h_kernel_alone():
    with torch.cuda.stream(cs):
        ops.copy_to_pinned(s.h_masks, s.d_masks, max_blocks=32)
    cs.synchronize()
print("copy kernel alone (32 blocks)      : %.2f ms" % timeit(d2h_kernel_alone, 10))
ops.paste_bits(e.masks128, e.boxes, 1024, 1024, out=s.d_masks); torch.cuda.synchronize()
ops.copy_to_pinned(s.h_masks, s.d_masks); torch.cuda.synchronize()
print("copy correct:", bool((s.h_masks == s.d_masks.cpu()).all()))

# same experiments with the compute work on a NON-default stream
comp = torch.cuda.Stream()
with torch.cuda.stream(comp):
    print("[side stream] replay only                   : %.2f ms" % timeit(lambda: e.graph.replay()))
    print("[side stream] + events + D2H on copy stream : %.2f ms" % timeit(c))
    def c3():
        b()
        cur = torch.cuda.current_stream()
        ev1.record(cur)
        with torch.cuda.stream(cs):
            cs.wait_event(ev1)
            s.h_rec.copy_(s.d_rec, non_blocking=True)
            ops.copy_to_pinned(s.h_masks, s.d_masks, max_blocks=16)
            ev2.record(cs)
    print("[side stream] + D2H by a 16-block copy kernel: %.2f ms" % timeit(c3))

# D2H of the PREVIOUS image enqueued AFTER the next replay has been launched
slots = e.slots
state = {"i": 0, "pending": None}
evc = [torch.cuda.Event(), torch.cuda.Event()]
evd = [torch.cuda.Event(), torch.cuda.Event()]
def late(kernel_copy):
    def f():
        i = state["i"]; sl = slots[i & 1]
        cur = torch.cuda.current_stream()
        e.image.copy_(image, non_blocking=True); e.graph.replay()
        # previous image's transfer starts now, behind the launch
        if state["pending"] is not None:
            j = state["pending"]; sp = slots[j & 1]
            with torch.cuda.stream(cs):
                cs.wait_event(evc[j & 1])
                sp.h_rec.copy_(sp.d_rec, non_blocking=True)
                if kernel_copy: ops.copy_to_pinned(sp.h_masks, sp.d_masks, max_blocks=16)
                else: sp.h_masks.copy_(sp.d_masks, non_blocking=True)
                evd[j & 1].record(cs)
        cur.wait_event(evd[i & 1])           # slot reuse
        sl.d_rec.copy_(e.rec, non_blocking=True)
        ops.paste_bits(e.masks128, e.boxes, 1024, 1024, out=sl.d_masks)
        evc[i & 1].record(cur)
        state["pending"] = i; state["i"] = i + 1
    return f
print("late D2H (memcpy)  after next launch : %.2f ms" % timeit(late(False)))
print("late D2H (kernel)  after next launch : %.2f ms" % timeit(late(True)))
