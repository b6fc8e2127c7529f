#!/usr/bin/env python3
"""Phase picture of a pooled job from a rocprofv3 kernel trace: per time bin, the fraction of the bin with >= 1 conv kernel
on the chip, the mean number of conv / coder kernels in flight, and per coder kernel class its mean duration and the mean
gap to the previous launch of the same stream (queue).  Usage: phase_timeline.py <rocprof dir> [bin_ms=20] [last_ms=1500]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
last_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 1500.0
rows = list(csv.DictReader(open(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    k = "conv" if "conv_mfma" in n else ("dec" if "rans_decode" in n else ("enc" if "rans_encode" in n else "other"))
    ev.append((s, e, k, r.get("Queue_Id", "0")))
ev.sort()
t_end = max(e for _, e, _, _ in ev)
# the job's timed region is the last dense stretch before the single-instance passes: take a window ending where the
# number of distinct queues in flight drops; simple and robust enough: the caller gives the window length and an offset
off_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
w1 = t_end - int(off_ms * 1e6)
w0 = w1 - int(last_ms * 1e6)
nb = int(last_ms / bin_ms)
busy = defaultdict(lambda: [0.0] * nb)
for s, e, k, q in ev:
    if e <= w0 or s >= w1:
        continue
    s, e = max(s, w0), min(e, w1)
    b0, b1 = int((s - w0) / 1e6 / bin_ms), min(nb - 1, int((e - w0) / 1e6 / bin_ms))
    for b in range(b0, b1 + 1):
        lo, hi = w0 + b * bin_ms * 1e6, w0 + (b + 1) * bin_ms * 1e6
        busy[k][b] += max(0.0, min(e, hi) - max(s, lo)) / (bin_ms * 1e6)
print(f"window: last {last_ms:.0f} ms (ending {off_ms:.0f} ms before the trace's end), bins of {bin_ms:.0f} ms: mean kernels in flight")
print("  t(ms)   conv    dec    enc  other")
for b in range(nb):
    print(f"{b * bin_ms:7.0f} {busy['conv'][b]:6.2f} {busy['dec'][b]:6.2f} {busy['enc'][b]:6.2f} {busy['other'][b]:6.2f}")
for kind in ("dec", "enc"):
    byq = defaultdict(list)
    for s, e, k, q in ev:
        if k == kind and s >= w0 and e <= w1:
            byq[q].append((s, e))
    durs = [e - s for v in byq.values() for s, e in v]
    gaps = [v[i + 1][0] - v[i][1] for v in byq.values() for i in range(len(v) - 1)]
    if durs:
        gaps.sort()
        print(f"{kind}: {len(durs)} launches, mean duration {sum(durs)/len(durs)/1e3:.1f} us, median gap to the queue's previous "
              f"{kind} launch {gaps[len(gaps)//2]/1e3 if gaps else 0:.1f} us")
