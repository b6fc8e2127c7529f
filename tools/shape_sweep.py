#!/usr/bin/env python3
"""Every tile configuration of the conv kernel on ONE layer shape (kernel-only timing, isolated launches):
shape_sweep.py n cin h w cout k [splitk]   -- prints the configurations sorted by time."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgbd_amd  # noqa: E402,F401
from rgbd_amd._lib import lib  # noqa: E402

n, cin, h, w, cout, k = (int(v) for v in sys.argv[1:7])
split = int(sys.argv[7]) if len(sys.argv) > 7 else 0
L = lib()
L.rgbd_debug_force_splitk(split)
res = []
cfgs = [(2, mt, nt) for mt in range(1, 6) for nt in (1, 2, 4)] + [(2, mt, 8) for mt in (1, 2, 3)] + \
       [(1, mt, nt) for mt in (1, 2, 3) for nt in (1, 2, 4)]
for wm, mt, nt in cfgs:
    for kc, dma in ((16, 0), (16, 1), (16, 2), (16, 3), (64, 0)):
        cfg = f"{wm},{mt},{nt},{kc},{dma}"
        L.rgbd_debug_force_tile(cfg.encode())
        ms = ctypes.c_float(0)
        rc = L.rgbd_conv_bench(n, cin, h, w, cout, k, 1, k // 2, 0, 0, 5, ctypes.byref(ms))
        if rc == 0:
            res.append((ms.value * 1e3, cfg))
L.rgbd_debug_force_tile(b"")
ms = ctypes.c_float(0)
L.rgbd_conv_bench(n, cin, h, w, cout, k, 1, k // 2, 0, 0, 5, ctypes.byref(ms))
gf = 2.0 * n * h * w * cout * cin * k * k / 1e9
print(f"shape n={n} cin={cin} {h}x{w} cout={cout} k={k} splitk={split}: table/cost-model pick {ms.value*1e3:.1f} us = {gf/ms.value:.1f} TF/s")
for us, cfg in sorted(res)[:12]:
    print(f"   {cfg:14s} {us:8.1f} us  {gf/us*1e3:6.1f} TF/s")
