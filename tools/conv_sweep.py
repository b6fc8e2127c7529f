#!/usr/bin/env python3
"""Kernel-only timing of the conv shapes that dominate ELIC_united (B images of HxW).  Usage: conv_sweep.py [B H W]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgbd_amd  # noqa: E402
from rgbd_amd._lib import _SO  # noqa: E402

L = ctypes.CDLL(_SO)
L.rgbd_conv_bench.restype = ctypes.c_int
L.rgbd_conv_bench.argtypes = [ctypes.c_int32] * 11 + [ctypes.POINTER(ctypes.c_float)]
B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 256, 256)
SPLIT = int(os.environ.get("SWEEP_SPLITK", "0"))  # -1: the codec's automatic split for entropy-model layers
h2, w2, h4, w4, h8, w8, h16, w16 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8, H // 16, W // 16
# name, cin, h, w, cout, k, stride, pad, transposed, residual, count per enc+dec
S = [
    ("s2 3x3 96->96", 96, h2, w2, 96, 3, 1, 1, 0, 0, 12), ("s2 1x1 192->96", 192, h2, w2, 96, 1, 1, 0, 0, 0, 10),
    ("s2 1x1 96->192 +res", 96, h2, w2, 192, 1, 1, 0, 0, 1, 12), ("s2 3x3 192->96 (spf)", 192, h2, w2, 96, 3, 1, 1, 0, 0, 4),
    ("s2 1x1 384->192", 384, h2, w2, 192, 1, 1, 0, 0, 0, 4), ("s2 1x1 192->48 (esa)", 192, h2, w2, 48, 1, 1, 0, 0, 0, 4),
    ("s1->s2 5x5s2 16->192", 3, H, W, 192, 5, 2, 2, 0, 0, 2), ("s2->s4 5x5s2 384->192", 384, h2, w2, 192, 5, 2, 2, 0, 0, 2),
    ("s4 3x3 96->96", 96, h4, w4, 96, 3, 1, 1, 0, 0, 24), ("s4 1x1 192->96", 192, h4, w4, 96, 1, 1, 0, 0, 0, 24),
    ("s4->s8 5x5s2 384->192", 384, h4, w4, 192, 5, 2, 2, 0, 0, 2), ("s8 3x3 96->96", 96, h8, w8, 96, 3, 1, 1, 0, 0, 12),
    ("s8->s16 5x5s2 384->320", 384, h8, w8, 320, 5, 2, 2, 0, 0, 2), ("s16 3x3 160->160", 160, h16, w16, 160, 3, 1, 1, 0, 0, 24),
    ("s16 1x1 320->160", 320, h16, w16, 160, 1, 1, 0, 0, 0, 24),
    ("deconv s16->s8 320->192", 320, h16, w16, 192, 5, 2, 2, 1, 0, 2), ("deconv s8->s4 192->192", 192, h8, w8, 192, 5, 2, 2, 1, 0, 2),
    ("deconv s4->s2 192->192", 192, h4, w4, 192, 5, 2, 2, 1, 0, 2), ("deconv s2->s1 192->3", 192, h2, w2, 3, 5, 2, 2, 1, 0, 2),
    ("hs deconv 384->320", 384, h16 // 4, w16 // 4, 320, 5, 2, 2, 1, 0, 4), ("hs deconv 640->480", 640, h16 // 2, w16 // 2, 480, 5, 2, 2, 1, 0, 4),
    ("hs deconv3 960->640", 960, h16, w16, 640, 3, 1, 1, 1, 0, 4),
    ("ep 1x1 1280->213", 1280, h16, w16, 213, 1, 1, 0, 0, 0, 2), ("ep 1x1 2816->469", 2816, h16, w16, 469, 1, 1, 0, 0, 0, 4),
    ("ep 3x3 469->512", 469, h16, w16, 512, 3, 1, 1, 0, 0, 8), ("ep 5x5 512->384", 512, h16, w16, 384, 5, 1, 2, 0, 0, 8),
    ("ep 3x3 213->42", 213, h16, w16, 42, 3, 1, 1, 0, 0, 8), ("ep 5x5 42->32", 42, h16, w16, 32, 5, 1, 2, 0, 0, 8),
    ("ep 5x5 170->128", 170, h16, w16, 128, 5, 1, 2, 0, 0, 8),
    ("chctx 5x5 128->224", 128, h16, w16, 224, 5, 1, 2, 0, 0, 4), ("chctx 5x5 224->128", 224, h16, w16, 128, 5, 1, 2, 0, 0, 16),
    ("chctx 5x5 128->384", 128, h16, w16, 384, 5, 1, 2, 0, 0, 4), ("locctx 5x5 192->384", 192, h16, w16, 384, 5, 1, 2, 0, 0, 6),
]
tot_ms = tot_gf = 0.0
for name, cin, h, w, cout, k, s, p, tr, res, cnt in S:
    ms = ctypes.c_float(0)
    L.rgbd_debug_force_splitk(SPLIT if name.split()[0] in ("ep", "chctx", "locctx", "hs") else 0)
    rc = L.rgbd_conv_bench(B, cin, h, w, cout, k, s, p, tr, res, 5, ctypes.byref(ms))
    if rc:
        print(name, "failed", rc)
        continue
    oh, ow = (h * s, w * s) if tr else ((h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1)
    gf = 2.0 * B * oh * ow * cout * cin * k * k / (s * s if tr else 1) / 1e9
    tot_ms += ms.value * cnt
    tot_gf += gf * cnt
    print(f"{name:28s} {ms.value*1e3:9.1f} us {gf:8.2f} GF {gf/ms.value:8.1f} TF/s  x{cnt:2d} = {ms.value*cnt:7.2f} ms")
print(f"weighted total {tot_ms:.1f} ms, {tot_gf:.0f} GF, {tot_gf/tot_ms:.1f} TF/s")
