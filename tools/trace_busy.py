"""GPU busy analysis of a rocprofv3 kernel trace (csv): union of kernel intervals vs wall time, average concurrency, and the
kernels that account for the time in which only ONE kernel is resident (the serial, latency-bound stretches)."""
import csv
import sys
from collections import defaultdict


def main(path, skip_frac=0.4):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # steady state = the `span` ms window with the largest summed kernel time (model setup, graph capture and the instrumented
    # eager passes around the timed region are sparser)
    span, binw = int(skip_frac * 1e6) if skip_frac > 1 else 200_000_000, 5_000_000
    t0 = rows[0][0]
    nb = (rows[-1][1] - t0) // binw + 2
    bins = [0] * nb
    for s_, e_, _ in rows:
        bins[(s_ - t0) // binw] += e_ - s_
    k = span // binw
    best, acc, lo = -1, sum(bins[:k]), 0
    for i in range(0, nb - k):
        if acc > best:
            best, lo = acc, i
        acc += bins[i + k] - bins[i]
    w0, w1 = t0 + lo * binw, t0 + (lo + k) * binw
    rows = [r for r in rows if r[0] >= w0 and r[1] <= w1]
    wall = rows[-1][1] - rows[0][0]
    events = []
    for s, e, n in rows:
        events.append((s, 1, n))
        events.append((e, -1, n))
    events.sort()
    busy = 0
    conc_time = defaultdict(int)
    solo = defaultdict(int)
    active = {}
    last = events[0][0]
    depth = 0
    for t, d, n in events:
        if depth > 0:
            busy += t - last
            conc_time[min(depth, 8)] += t - last
            if depth == 1:
                solo[next(iter(active))] += t - last
        last = t
        depth += d
        if d > 0:
            active[n] = active.get(n, 0) + 1
        else:
            active[n] -= 1
            if active[n] == 0:
                del active[n]
    total_k = sum(e - s for s, e, _ in rows)
    print(f"window {wall / 1e6:.1f} ms, {len(rows)} kernels, busy {busy / wall:.3f}, sum of kernel durations / wall = {total_k / wall:.2f}")
    print("time share by number of resident kernels:", {k: round(v / wall, 3) for k, v in sorted(conc_time.items())})
    print("top kernels running ALONE (ms):")
    for n, v in sorted(solo.items(), key=lambda kv: -kv[1])[:14]:
        print(f"  {v / 1e6:8.2f}  {n[:110]}")
    by = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        by[n][0] += 1
        by[n][1] += e - s
    print("top kernels by summed duration (ms, calls):")
    for n, (c, v) in sorted(by.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"  {v / 1e6:8.2f} {c:6d}  {n[:110]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)
