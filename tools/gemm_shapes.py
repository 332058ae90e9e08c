"""Per-launch-shape GEMM durations from a rocprofv3 kernel trace (grid size as the shape proxy)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 28
f = glob.glob(d + '/*/*kernel_trace.csv')[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gemm_bf16' in r['Kernel_Name']:
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        g[(name, int(r['Grid_Size_X']) // 256)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:36s} blocks={k[1]:6d} calls/img={len(v)/n_img:5.1f} avg={sum(v)/len(v):8.1f}us  ms/img={sum(v)/1e3/n_img:6.3f}")
    tot += sum(v) / 1e3 / n_img
print("total gemm ms/img", round(tot, 3))
