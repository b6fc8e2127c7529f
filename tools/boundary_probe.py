#!/usr/bin/env python3
"""Do tiny kernels on OTHER streams slow a large convolution down?  (in a pooled job the transforms of one cohort
run 1.5x slower while another cohort is in its coder phase, which is hundreds of small launches.)  Every kernel boundary is
an agent-scope release / acquire -- on this eight-XCD part an L2 write-back / invalidate -- so a stream of tiny kernels may
keep evicting what the big kernels share in L2 (weights).  This probe times one conv layer (kernel-only, back to back on the
NULL stream) alone and while T host threads each push tiny kernels through their own stream as fast as they can.

    python tools/boundary_probe.py [threads ...]        (default 0 1 4 10)
"""
import ctypes
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch  # noqa: E402

import rgbd_amd  # noqa: E402,F401
from rgbd_amd._lib import lib  # noqa: E402

L = lib()
SHAPES = [(8, 192, 128, 160, 192, 3, 1), (8, 384, 256, 320, 192, 5, 2), (8, 96, 256, 320, 96, 3, 1),
          (8, 192, 64, 80, 96, 1, 1), (4, 96, 32, 40, 96, 3, 1)]  # the last two: 20-60 us launches (dispatch latency shows)
threads = [int(v) for v in sys.argv[1:]] or [0, 1, 4, 10]


def conv_ms(shape, iters=40):
    n, cin, h, w, cout, k, s = shape
    ms = ctypes.c_float(0)
    rc = L.rgbd_conv_bench(n, cin, h, w, cout, k, s, k // 2, 0, 0, iters, ctypes.byref(ms))
    assert rc == 0, rc
    return ms.value


stop = threading.Event()
counts = []


def pest(i, kind):
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    x = torch.zeros(64 if kind == "tiny" else 1 << 20, device="cuda")
    n = 0
    with torch.cuda.stream(st):
        while not stop.is_set():
            for _ in range(64):
                x.add_(1.0)
            n += 64
            st.synchronize()
    counts.append(n)


for kind in ("tiny", "4MB"):
    for T in threads:
        stop.clear()
        counts.clear()
        ts = [threading.Thread(target=pest, args=(i, kind)) for i in range(T)]
        for t in ts:
            t.start()
        time.sleep(0.3)
        t0 = time.time()
        res = [min(conv_ms(s), conv_ms(s)) for s in SHAPES]
        dt = time.time() - t0
        stop.set()
        for t in ts:
            t.join()
        rate = sum(counts) / max(dt + 0.3, 1e-6) / 1e3 if T else 0.0
        print(f"{kind:5s} kernels on {T:2d} other streams (~{rate:6.0f} k launches/s): conv ms " +
              "  ".join(f"{m:.3f}" for m in res), flush=True)
