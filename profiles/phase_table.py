#!/usr/bin/env python3
"""Phase table of the bench's timed round from a rocprofv3 kernel trace of the driver's command: how
many ms of the round have >= 10 conv kernels in flight, 1-10, only coder kernels, or nothing -- so that
`roofline.frac` = conv FLOPs / (time of the conv phases) can be checked against the per-kernel profile.

    phase_table.py <rocprof dir> <stats.log with the bench's JSON line> [bin_ms=2]

The timed round is located from the trace itself: the pooled part of the run is where most of the engine instances' hardware
queues (>= 8, or all but one of fewer instances) have kernels in flight; its last `ms_per_step x steps` milliseconds (from the JSON line) are the timed region (the warm-up rounds precede it,
the single-instance conv passes follow it)."""
import csv
import glob
import json
import sys
from collections import defaultdict

d, log = sys.argv[1], sys.argv[2]
bin_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
line = json.loads([ln for ln in open(log) if ln.startswith("{")][-1])
dur_ms = line["ms_per_step"] * line["steps"]
gflop_step = line["roofline"]["gflop_per_step"]
rows = list(csv.DictReader(open(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    k = "conv" if ("conv_mfma" in n or "splitk_reduce" in n) else ("coder" if "rans_" in n else "other")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "0")))
ev.sort()
t0 = ev[0][0]
nb = int((ev[-1][1] - t0) / 1e6 / bin_ms) + 1
queues = [set() for _ in range(nb)]
fl = {k: [0.0] * nb for k in ("conv", "coder", "other")}
for s, e, k, q in ev:
    b0, b1 = int((s - t0) / 1e6 / bin_ms), int((e - t0) / 1e6 / bin_ms)
    for b in range(b0, min(b1, nb - 1) + 1):
        lo, hi = t0 + b * bin_ms * 1e6, t0 + (b + 1) * bin_ms * 1e6
        fl[k][b] += max(0.0, min(e, hi) - max(s, lo)) / (bin_ms * 1e6)
        queues[b].add(q)
# end of the pooled part: the last bin of a 40 ms window in which >= 8 queues were active
win = max(1, int(40 / bin_ms))
inst = int(line["config"]["engine_instances"])
need_q = min(8, max(2, inst - 1))
end = max(b for b in range(nb) if len(set().union(*queues[max(0, b - win):b + 1])) >= need_q)
many = 10 if inst >= 16 else max(2, inst // 2)  # "most instances are in a transform phase": 10 of 20, 2 of 5
start = max(0, end - int(dur_ms / bin_ms) + 1)
cls = defaultdict(float)
K_MANY, K_FEW = f">= {many} conv kernels in flight", f"1 - {many} conv kernels in flight"
for b in range(start, end + 1):
    c, r_ = fl["conv"][b], fl["coder"][b]
    key = K_MANY if c >= many else (K_FEW if c >= 1 else
                                                         ("coder kernels only (< 1 conv)" if r_ >= 0.5 else "neither (transitions, host)"))
    cls[key] += bin_ms
span = (end - start + 1) * bin_ms
print(f"timed round: {span:.0f} ms of trace ({line['steps']} steps x {line['ms_per_step']} ms = {dur_ms:.0f} ms by the bench's clock), "
      f"{line['config']['engine_instances']} engine instances, bins of {bin_ms:g} ms")
for k in (K_MANY, K_FEW, "coder kernels only (< 1 conv)", "neither (transitions, host)"):
    print(f"  {k:34s} {cls[k]:8.0f} ms  {100 * cls[k] / span:5.1f} %")
conv_ms = cls[K_MANY] + cls[K_FEW]
tf = gflop_step * line["steps"] / conv_ms if conv_ms else 0.0
print(f"conv FLOPs of the round / time with conv kernels in flight: {gflop_step * line['steps'] / 1e3:.1f} TFLOP / {conv_ms:.0f} ms = "
      f"{tf:.1f} TFLOP/s = {tf / 157.3:.3f} of the fp32 MFMA peak; / the whole round: "
      f"{gflop_step * line['steps'] / span:.1f} TFLOP/s = {gflop_step * line['steps'] / span / 157.3:.3f} (the line's roofline.frac: {line['roofline']['frac']})")
