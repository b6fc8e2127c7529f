#!/usr/bin/env python3
"""GPU timeline of the B = 1 calling pattern from a rocprofv3 kernel trace of tools/latency_b1.py: for each steady-state
iteration (delimited by the long y-stream encode launches: one decompress() + the next compress()) the span, the time some
kernel is running, the coder / conv / other shares and the largest idle gaps (host work between launches shows up as gaps).
Usage: rocprofv3 --kernel-trace --output-format csv -d out -o p -- python3 tools/latency_b1.py 512 640
       python3 tools/latency_gaps.py out"""
import csv
import glob
import sys

rows = sorted(csv.DictReader(open(glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0])),
              key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows]
encs = [i for i, e in enumerate(ev) if "rans_encode" in e[2] and e[1] - e[0] > 10_000_000]
for a, b in zip(encs[-4:-1], encs[-3:]):
    seg = ev[a + 1:b + 1]
    t0 = end = ev[a][1]
    busy, gaps = 0, []
    for s, e, n in seg:
        if s > end:
            gaps.append((s - end, n, (end - t0) / 1e6))
        busy += max(0, e - max(s, end))
        end = max(end, e)
    coder = sum(e - s for s, e, n in seg if "rans_" in n)
    conv = sum(e - s for s, e, n in seg if "conv_mfma" in n)
    print(f"iteration: {len(seg)} kernels, span {(end - t0) / 1e6:.2f} ms, some kernel running {busy / 1e6:.2f}, idle "
          f"{(end - t0 - busy) / 1e6:.2f}; coder {coder / 1e6:.2f}, conv {conv / 1e6:.2f}, other {(busy - coder - conv) / 1e6:.2f} ms")
    for g, n, at in sorted(gaps, reverse=True)[:4]:
        print(f"     gap {g / 1e3:8.1f} us at +{at:7.2f} ms before {n}")
