"""Do independent branches inside one hipGraph run concurrently?  40 small GEMMs sequential vs fork/join over 2 and 4 streams."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ape_amd.ops as ops
bf = torch.bfloat16
def mk(M, N, K):
    return torch.randn(M, K, device="cuda").to(bf), (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf), torch.empty(M, N, device="cuda", dtype=bf)
def run(shape, nbranch, per_branch=10):
    sets = [[mk(*shape) for _ in range(per_branch)] for _ in range(nbranch)]
    streams = [torch.cuda.Stream() for _ in range(nbranch)]
    def body():
        cur = torch.cuda.current_stream()
        if nbranch == 1:
            for a, w, o in sets[0]: ops.gemm(a, w, None, out=o)
            return
        ev0 = torch.cuda.Event(); ev0.record(cur)
        evs = []
        for s, st in zip(streams, sets):
            s.wait_event(ev0)
            with torch.cuda.stream(s):
                for a, w, o in st: ops.gemm(a, w, None, out=o)
                e = torch.cuda.Event(); e.record(s); evs.append(e)
        for e in evs: cur.wait_event(e)
    for _ in range(2): body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 20 * 1e3
for shape in [(900, 256, 256), (900, 2048, 256), (4096, 1024, 1024)]:
    t1 = run(shape, 1, 40); t2 = run(shape, 2, 20); t4 = run(shape, 4, 10)
    print(f"{shape}: 40 GEMMs sequential {t1:.0f} us | 2 branches x 20: {t2:.0f} us | 4 branches x 10: {t4:.0f} us", flush=True)
