#!/usr/bin/env python3
"""For each Bi-CEE / h_s layer shape at one latent size: time every tile shape x KC x staging mode x split-K and compare the
best with what the launcher's cost model picks.  Usage: tile_sweep.py [B h w]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgbd_amd  # noqa: E402,F401
from rgbd_amd._lib import lib  # noqa: E402

L = lib()
B, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 16, 16)
S = [("ep 1x1 1280->213", 1280, 213, 1), ("ep 1x1 1664->277", 1664, 277, 1), ("ep 1x1 2816->469", 2816, 469, 1),
     ("ep 3x3 213->42", 213, 42, 3), ("ep 3x3 469->512", 469, 512, 3), ("ep 5x5 512->384", 512, 384, 5),
     ("ep 5x5 170->128", 170, 128, 5), ("ep 5x5 42->32", 42, 32, 5),
     ("chctx 5x5 128->224", 128, 224, 5), ("chctx 5x5 224->128", 224, 128, 5), ("chctx 5x5 128->384", 128, 384, 5),
     ("locctx 5x5 192->384", 192, 384, 5), ("locctx 5x5 64->128", 64, 128, 5), ("attn 3x3 160->160", 160, 160, 3),
     ("attn 1x1 320->160", 320, 160, 1), ("attn 1x1 160->320", 160, 320, 1)]
TILES = ([(2, m, 8) for m in (3, 2, 1)] + [(2, m, n) for n in (4, 2, 1) for m in (5, 4, 3, 2, 1)] +
         [(1, m, n) for n in (4, 2, 1) for m in (3, 2, 1)])


def run(cin, cout, k, iters=4):
    ms = ctypes.c_float(0)
    rc = L.rgbd_conv_bench(B, cin, h, w, cout, k, 1, k // 2, 0, 0, iters, ctypes.byref(ms))
    return ms.value if rc == 0 else float("inf")


for name, cin, cout, k in S:
    gf = 2.0 * B * h * w * cout * cin * k * k / 1e9
    L.rgbd_debug_force_tile(b"")
    L.rgbd_debug_force_splitk(-1)
    auto = run(cin, cout, k)
    res = []
    for wm, mt, nt in TILES:
        for kc, dma in ((16, 1), (16, 0), (64, 0)):
            for sk in (1, 2, 4, 8):
                L.rgbd_debug_force_tile(f"{wm},{mt},{nt},{kc},{dma}".encode())
                L.rgbd_debug_force_splitk(sk)
                res.append((run(cin, cout, k, 3), wm, mt, nt, kc, dma, sk))
    res.sort()
    best = res[0]
    per_s = {sk: min(r for r in res if r[6] == sk) for sk in (1, 2, 4, 8)}
    auto_s = {}
    for sk in (1, 2, 4, 8):
        L.rgbd_debug_force_tile(b"")
        L.rgbd_debug_force_splitk(sk)
        auto_s[sk] = run(cin, cout, k, 3)
    print(f"    per split-K best/auto-tile us: " + "  ".join(
        f"s{sk}: {per_s[sk][0]*1e3:.0f}({per_s[sk][1]},{per_s[sk][2]},{per_s[sk][3]},{per_s[sk][4]},{per_s[sk][5]})/{auto_s[sk]*1e3:.0f}"
        for sk in (1, 2, 4, 8)))
    print(f"{name:22s} auto {auto*1e3:7.1f} us {gf/auto:6.1f} TF/s | best {best[0]*1e3:7.1f} us {gf/best[0]:6.1f} TF/s "
          f"cfg wm={best[1]} mt={best[2]} nt={best[3]} kc={best[4]} dma={best[5]} splitk={best[6]} | next "
          + " ".join(f"{r[0]*1e3:.0f}us({r[1]},{r[2]},{r[3]},{r[4]},{r[5]},s{r[6]})" for r in res[1:4]), flush=True)
