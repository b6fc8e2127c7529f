#!/usr/bin/env python3
"""Bit-exactness of forced tile variants against the default choice (the determinism contract: tile shape never changes a
result).  python tools/force_check.py "2,3,8,16,1" ["2,2,8,64,0" ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rgbd_amd  # noqa: E402,F401
from rgbd_amd._lib import lib  # noqa: E402

L = lib()
f32p = ctypes.POINTER(ctypes.c_float)
SHAPES = [  # n cin h w cout k stride pad transposed
    (2, 96, 32, 48, 96, 3, 1, 1, 0), (1, 192, 40, 24, 96, 1, 1, 0, 0), (2, 96, 16, 16, 192, 1, 1, 0, 0),
    (1, 64, 33, 47, 80, 3, 1, 1, 0), (2, 48, 32, 32, 64, 5, 1, 2, 0), (1, 96, 64, 64, 96, 5, 2, 2, 0),
    (1, 64, 16, 24, 96, 5, 2, 2, 1), (1, 16, 64, 64, 32, 5, 2, 2, 0), (1, 96, 20, 36, 3, 3, 1, 1, 0)]


def run(shape, x, w, b):
    n, cin, h, wd, cout, k, s, p, tr = shape
    oh = (h - 1) * s - 2 * p + k + (s - 1) if tr else (h + 2 * p - k) // s + 1
    ow = (wd - 1) * s - 2 * p + k + (s - 1) if tr else (wd + 2 * p - k) // s + 1
    y = torch.empty((n, cout, oh, ow), device="cuda")
    rc = L.rgbd_conv2d_nchw(ctypes.c_void_p(x.data_ptr()), n, cin, h, wd, w.numpy().ctypes.data_as(f32p),
                            b.numpy().ctypes.data_as(f32p), cout, k, s, p, tr, 1, None, ctypes.c_void_p(y.data_ptr()), None)
    return rc, y.cpu()


bad = 0
for cfg in sys.argv[1:]:
    nok = nskip = 0
    for shape in SHAPES:
        n, cin, h, wd, cout, k, s, p, tr = shape
        g = torch.Generator().manual_seed(sum(shape))
        x = torch.randn(n, cin, h, wd, generator=g).cuda()
        w = (torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) / (k * cin ** 0.5)).contiguous()
        b = torch.randn(cout, generator=g)
        L.rgbd_debug_force_tile(b"")
        rc0, y0 = run(shape, x, w, b)
        assert rc0 == 0
        L.rgbd_debug_force_tile(cfg.encode())
        rc1, y1 = run(shape, x, w, b)
        L.rgbd_debug_force_tile(b"")
        if rc1 != 0:
            nskip += 1
            continue
        if not torch.equal(y0, y1):
            bad += 1
            print(f"MISMATCH {cfg} {shape}: max diff {(y0 - y1).abs().max().item():.3e}")
        else:
            nok += 1
    print(f"{cfg}: {nok} shapes bit-identical, {nskip} not launchable with this tile")
sys.exit(1 if bad else 0)
