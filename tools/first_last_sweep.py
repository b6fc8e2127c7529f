#!/usr/bin/env python3
"""Tile sweep for the two odd layers of the transforms: the first conv (3 or 1 -> 192, 5x5 stride 2 on the full image) and
the last transposed conv (192 -> 3 or 1).  Usage: first_last_sweep.py B H W"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgbd_amd  # noqa: E402,F401
from rgbd_amd._lib import lib  # noqa: E402

L = lib()
B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (4, 512, 640)
TILES = ([(2, m, 8) for m in (3, 2, 1)] + [(2, m, n) for n in (4, 2, 1) for m in (5, 4, 3, 2, 1)] +
         [(1, m, n) for n in (4, 2, 1) for m in (3, 2, 1)])


def run(cin, h, w, cout, transposed, iters=3):
    ms = ctypes.c_float(0)
    rc = L.rgbd_conv_bench(B, cin, h, w, cout, 5, 2, 2, transposed, 0, iters, ctypes.byref(ms))
    return ms.value if rc == 0 else float("inf")


for name, cin, h, w, cout, tr in (("first conv rgb 3->192", 3, H, W, 192, 0), ("first conv depth 1->192", 1, H, W, 192, 0),
                                  ("last deconv 192->3", 192, H // 2, W // 2, 3, 1), ("last deconv 192->1", 192, H // 2, W // 2, 1, 1)):
    L.rgbd_debug_force_tile(b"")
    auto = run(cin, h, w, cout, tr)
    res = []
    for wm, mt, nt in TILES:
        for kc, dma in ((16, 1), (16, 0), (16, 2), (16, 3)):
            L.rgbd_debug_force_tile(f"{wm},{mt},{nt},{kc},{dma}".encode())
            res.append((run(cin, h, w, cout, tr, 2), wm, mt, nt, kc, dma))
    res.sort()
    print(f"{name:26s} auto {auto*1e3:8.1f} us | best " + " ".join(f"{r[0]*1e3:.0f}us({r[1]},{r[2]},{r[3]},{r[4]},{r[5]})" for r in res[:5]), flush=True)
L.rgbd_debug_force_tile(b"")
