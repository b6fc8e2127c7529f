#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals and conv launches grouped by grid."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
stats = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
trace = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
print(f"== per-kernel totals (divide by {steps:g} steps) ==")
for r in rows[:22]:
    print(r["Name"][:62].ljust(62), r["Calls"].rjust(6), f"{float(r['TotalDurationNs'])/1e6/steps:9.3f} ms/step",
          f"{float(r['AverageNs'])/1e3:9.1f} us avg", r["Percentage"].rjust(7), "%")
rows = list(csv.DictReader(open(trace)))
g = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if "conv_mfma" not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][22:42], int(r["Grid_Size_X"]) // 256, r["Grid_Size_Y"], r["Grid_Size_Z"], r["LDS_Block_Size"])
    g[key][0] += 1
    g[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in g.values())
print("== conv launches by (template, blocks.x, y, z, LDS) ==")
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1])[:30]:
    print(k, v[0], f"{v[1]/v[0]:8.1f} us avg", f"{100*v[1]/tot:5.1f} %")
ts = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
print("GPU busy ms", sum(e - s for s, e in ts) / 1e6, "span ms", (ts[-1][1] - ts[0][0]) / 1e6)
