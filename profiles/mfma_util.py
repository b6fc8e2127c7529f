#!/usr/bin/env python3
"""MFMA utilisation per kernel template from one rocprofv3 pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- <cmd>
(counters in their own run, kernel trace only for the durations).  Per dispatch:
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)
(the gfx94x MfmaUtil formula -- ROCm 7.2 has no gfx950 section for derived metrics, MI355X_MICROARCH.md "rocprofv3 PMC
slots"; SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs in cycles; GRBM_GUI_ACTIVE comes back summed over the chip's 8
XCDs -- per dispatch it is 8 x (duration x shader clock) -- hence the division by 8),
aggregated per kernel template weighted by busy cycles.  Counter collection serialises dispatches, so these are
isolated-launch figures.  Usage: mfma_util.py <rocprof_dir> <out.csv>"""
import collections
import csv
import glob
import re
import sys

d, out = sys.argv[1], sys.argv[2]
f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
disp = collections.defaultdict(dict)
name = {}
for r in csv.DictReader(open(f)):
    k = r["Dispatch_Id"]
    disp[k][r["Counter_Name"]] = disp[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    name[k] = r["Kernel_Name"]
dur = {}
tr = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
if tr:
    for r in csv.DictReader(open(tr[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])  # launches, mfma busy, gui active, cu busy, us
for k, c in disp.items():
    m = re.match(r"(void )?([A-Za-z_0-9]+(<[^>]*>)?)", name[k])
    key = m.group(2) if m else name[k][:50]
    a = agg[key]
    a[0] += 1
    a[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    a[2] += c.get("GRBM_GUI_ACTIVE", 0.0)
    a[3] += c.get("SQ_BUSY_CU_CYCLES", 0.0)
    a[4] += dur.get(k, 0.0)
tot = [sum(a[i] for key, a in agg.items() if "conv_mfma" in key) for i in range(5)]
with open(out, "w") as fo:
    fo.write("kernel,launches,total_us,mfma_busy_cycles,gui_active_cycles,mfma_util_pct\n")
    rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
    for key, a in rows:
        util = 100.0 * a[1] / (a[2] / 8 * 256 * 4) if a[2] else 0.0
        fo.write(f"\"{key}\",{a[0]},{a[4]:.1f},{a[1]:.0f},{a[2]:.0f},{util:.2f}\n")
    util = 100.0 * tot[1] / (tot[2] / 8 * 256 * 4) if tot[2] else 0.0
    fo.write(f"\"ALL conv_mfma_kernel\",{tot[0]},{tot[4]:.1f},{tot[1]:.0f},{tot[2]:.0f},{util:.2f}\n")
print(open(out).read())
