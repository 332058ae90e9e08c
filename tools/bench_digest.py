#!/usr/bin/env python
"""Digest of bench.py JSON lines: value, step time, the tile kernel's roofline (exact launch meter and event pair), by shape, and the
largest library kernels.  `--vs-rocprof <kernel_stats.csv> <line.json>`: the launch meter's per-kernel averages of an instrumented
pass next to rocprofv3's averages of the same launches (they must agree: both read the dispatch's begin / end timestamps)."""
import csv
import json
import re
import sys


def load(path):
    for line in open(path):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                pass
    return None


def canon(name):
    """a kernel name reduced to `base<numbers and booleans>`: comparable between a launch-site expression and a demangled symbol"""
    name = re.sub(r"^void ", "", name.strip().strip("()"))
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("unsigned short", "bf16_t").replace("_Float16", "f16_t")
    return re.sub(r"\s+", "", name)


def main():
    if sys.argv[1] == "--vs-rocprof":
        stats, line = sys.argv[2], load(sys.argv[3])
        roc = {}
        for row in csv.DictReader(open(stats)):
            roc[canon(row["Name"])] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
        lib = line["roofline"]["all_gemm_kernels"]["library_kernels"]
        print(f"{'launch-site expression':70s} {'meter us':>9s} {'rocprof us':>10s}  calls(meter/img)  rocprof calls")
        for k, (n, us, ms) in lib.items():
            c = canon(k)
            hit = roc.get(c) or next((v for kk, v in roc.items() if kk.startswith(c.split("<")[0] + "<") and kk.replace("bf16_t", "H").replace("f16_t", "H") == c.replace("bf16_t", "H")), None)
            print(f"{k[:70]:70s} {us:9.1f} {(hit[1] if hit else float('nan')):10.1f}  {n:8.2f}  {hit[0] if hit else '-'}")
        r = line["roofline"]
        print("dominant:", r["kernel"], "exact us", round(r["avg_launch_us"], 2), "event-pair us", round(r.get("avg_launch_us_event_pair", 0), 2), "frac", round(r["frac"], 4))
        return
    for path in sys.argv[1:]:
        d = load(path)
        if d is None:
            print(path, "no JSON line")
            continue
        r = d["roofline"]
        val = f"{d['value']:.2f} {d['unit']}  {d['ms_per_step']:.2f} ms/step" if d.get("value") else "(instrumented pass only: no timed region)"
        print(f"{path}: {val}  B={d['config'].get('images_per_step')}  dtype={d['dtype']}")
        print(f"  roofline {r['kernel']}: frac {r['frac']:.4f} ({r['achieved']:.0f} TF/s), {r['avg_launch_us']:.1f} us/launch exact"
              f" ({r.get('avg_launch_us_event_pair', float('nan')):.1f} event pair -> {r.get('frac_event_pair', float('nan')):.4f}), {r['launches_per_image']:.1f} launches/image, traffic {r.get('traffic')}")
        for sh, v in r.get("by_shape", {}).items():
            print(f"    {sh:45s} {v}")
        a = r["all_gemm_kernels"]
        print(f"  all GEMM kernels: {a['ms_per_image']:.2f} ms/image, {a['tflops']:.0f} TF/s")
        for k, v in list(a.get("by_kernel_ms_per_image", {}).items())[:8]:
            print(f"    {k:50s} {v:7.3f} ms/img  {a['by_kernel_tflops'][k]:7.1f} TF/s  {a['by_kernel_launches_per_image'][k]:6.1f} launches/img")
        for k, v in list(a.get("library_kernels", {}).items())[:14]:
            print(f"    lib {k[:80]:80s} {v}")


if __name__ == "__main__":
    main()
