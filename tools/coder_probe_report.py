#!/usr/bin/env python3
"""ns/symbol of the rANS kernels per probe class from a rocprofv3 kernel trace of tools/coder_probe.py.
Usage: coder_probe_report.py <rocprof_dir> <N>"""
import csv
import glob
import sys

d, N = sys.argv[1], float(sys.argv[2])
tr = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
enc = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "rans_encode" in r["Kernel_Name"]]
dec = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "rans_decode" in r["Kernel_Name"]]
names = ["idx0-7", "idx16-23", "idx32-39", "idx48-55", "idx0-63", "stress-like (sigma x3: escapes)", "zeros idx0-3"]
for i, (e, dd) in enumerate(zip(enc, dec)):
    print(f"{names[i] if i < len(names) else i:34s} encode {e / N:7.1f} ns/symbol   decode {dd / N:7.1f} ns/symbol")
