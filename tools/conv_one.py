#!/usr/bin/env python3
"""Runs one conv shape a few times (kernel-only) -- a target for rocprofv3 --pmc.  Args: n cin h w cout k stride transposed"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgbd_amd  # noqa: E402,F401
from rgbd_amd._lib import lib  # noqa: E402

n, cin, h, w, cout, k, s, tr = (int(v) for v in sys.argv[1:9])
ms = ctypes.c_float(0)
rc = lib().rgbd_conv_bench(n, cin, h, w, cout, k, s, k // 2, tr, 0, 6, ctypes.byref(ms))
print("rc", rc, "ms", ms.value, "TF/s", 2.0 * n * (h // s if not tr else h * s) * (w // s if not tr else w * s) * cout * cin * k * k / (s * s if tr else 1) / ms.value / 1e9)
