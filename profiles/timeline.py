#!/usr/bin/env python3
"""GPU occupancy of a rocprofv3 kernel trace: union of kernel intervals vs span, and time with >=1 conv kernel active."""
import csv
import glob
import sys

d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = "conv" if "conv_mfma" in r["Kernel_Name"] else ("rans" if "rans_" in r["Kernel_Name"] else "other")
    ev.append((s, e, k))
ev.sort()
t0 = ev[len(ev) // 3][0]  # skip warm-up third
ev = [x for x in ev if x[0] >= t0]
span = max(e for _, e, _ in ev) - t0


def union(kinds):
    iv = sorted((s, e) for s, e, k in ev if k in kinds)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        tot += ce - cs
    return tot


print(f"span {span/1e6:.1f} ms | any kernel {union({'conv','rans','other'})/1e6:.1f} | conv|other {union({'conv','other'})/1e6:.1f} | "
      f"conv {union({'conv'})/1e6:.1f} | rans {union({'rans'})/1e6:.1f} | sum conv durations {sum(e-s for s,e,k in ev if k=='conv')/1e6:.1f}")
