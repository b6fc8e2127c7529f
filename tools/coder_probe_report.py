#!/usr/bin/env python3
"""Prints ns/symbol per launch from a rocprofv3 kernel trace of tools/coder_probe.py."""
import csv
import glob
import sys

d, n = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    if "rans_" in r["Kernel_Name"]:
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        print(r["Kernel_Name"][:18], f"{ns/1e3:10.1f} us  {ns/n:7.1f} ns/symbol")
