"""Per-kernel means of the rocprofv3 PMC passes written by tools/gpu_pmc.sh (counter_collection.csv files)."""
import collections
import csv
import glob
import sys

import os

root = sys.argv[1]
# which build of the library these passes measured: bench.py refuses a `roofline.traffic` from a summary whose digest is not the
# digest of the sources it runs on (ape_amd/build.py _digest(): every csrc file + the header + the flags)
_stamp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ape_amd", "lib", "libape_hip.sha256")
print("# library_digest " + (open(_stamp).read().strip() if os.path.exists(_stamp) else "unknown"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for name, cs in agg.items():
    n = max(len(v) for v in cs.values())
    mean = {k: sum(v) / len(v) for k, v in cs.items()}
    rows.append((name, n, mean))
keys = sorted({k for _, _, m in rows for k in m})
# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the share of the busy CUs' SIMD cycles with the matrix pipe
# occupied (the gfx94x MfmaUtil formula; ROCm 7.2 ships no gfx950 derived counters).  It is a statement in CYCLES: at the clock the chip
# sustains under this load (~1.7 GHz against the 2.4 GHz behind the 2.5 PF/s peak) 60 % busy is ~1.05 PF/s.
print("kernel | launches | " + " | ".join(keys) + " | MFMA busy | HBM MB/launch = (2*FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction)")
for name, n, m in sorted(rows, key=lambda r: -r[2].get("FETCH_SIZE", 0) * r[1])[:25]:
    hbm = (2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) / 1024.0
    util = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4.0 * m["SQ_BUSY_CU_CYCLES"]) if m.get("SQ_BUSY_CU_CYCLES") else float("nan")
    print(f"{name[:60]:60s} | {n:5d} | " + " | ".join(f"{m.get(k, float('nan')):.4g}" for k in keys) + f" | {util:.3f} | {hbm:.1f}")
