"""Per-kernel means of the counter passes written by tools/gpu_pmc_msda.sh, with the derived ratios the MSDA question needs."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[name].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for name, cs in sorted(agg.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    if "msda" not in name and "gemm" not in name and "attn" not in name and "ffn" not in name:
        continue
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    us = sum(dur[name]) / max(len(dur[name]), 1) / 1e3 if name in dur else float("nan")
    print(f"== {name[:100]}  ({len(dur.get(name, []))} launches over all passes, avg {us:.1f} us under the profiler)")
    for k in sorted(m):
        print(f"   {k:40s} {m[k]:.6g}")
    g = m.get
    if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU") is not None:
        print(f"   -> VALU active / wave cycles            {g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'):.3f}")
    # VALU-busy fraction of the SIMDs' time.  SQ_ACTIVE_INST_VALU counts QUAD-cycles (4 shader cycles, MI355X_MICROARCH.md
    # "s_memtime tick vs SQ PMC units") summed over all waves; the chip offers 1024 SIMDs (256 CUs x 4) x GRBM_GUI_ACTIVE cycles
    # while the kernel runs (GRBM_GUI_ACTIVE is summed over the 8 XCDs by this rocprofv3: / 8):
    #     VALU busy = 4 * SQ_ACTIVE_INST_VALU / (1024 * GRBM_GUI_ACTIVE / 8)
    if g("SQ_ACTIVE_INST_VALU") is not None and g("GRBM_GUI_ACTIVE"):
        busy = 4.0 * g("SQ_ACTIVE_INST_VALU") / (1024.0 * g("GRBM_GUI_ACTIVE") / 8.0)
        print(f"   -> VALU busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)   {busy:.3f}")
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY") is not None:
        print(f"   -> issue stall (WAIT_INST_ANY) / cycles  {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}")
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
        print(f"   -> parked (WAIT_ANY) / cycles            {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}")
    if g("SQ_WAVES") and g("SQ_INSTS_VALU") is not None:
        print(f"   -> VALU instructions per wave            {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.1f}")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
        print(f"   -> L2 hit rate                           {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.4f}")
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
        print(f"   -> L1 (TCP) miss ratio: TCC read req / cache accesses   {g('TCP_TCC_READ_REQ_sum') / g('TCP_TOTAL_CACHE_ACCESSES_sum'):.4f}")
    if g("TA_TA_BUSY_sum") and g("GRBM_GUI_ACTIVE"):
        print(f"   -> TA busy (sum over TAs) / GPU active cycles  {g('TA_TA_BUSY_sum') / g('GRBM_GUI_ACTIVE'):.1f}  (256 TAs: /256 = {g('TA_TA_BUSY_sum') / g('GRBM_GUI_ACTIVE') / 256:.3f})")
    if g("FETCH_SIZE") is not None:
        print(f"   -> HBM-side read MB per launch (2 x FETCH_SIZE KB, gfx950 correction)  {2 * g('FETCH_SIZE') / 1024:.1f}")
