#!/usr/bin/env python3
"""What would one launch for both modalities buy?  The RGB and the depth branch of g_a / g_s / h_a / h_s / the channel-context
nets run the same layer shapes on independent data, so a layer pair can be one launch with twice the workgroups.  A paired
launch behaves like the same layer at twice the batch (same tiles, two weight sets): this probe times every conv shape of
one c3 compress()+decompress() kernel-only at N and at 2N and prints t(2N) / (2 t(N)) weighted by launch counts.

    python tools/pair_probe.py [B,H,W]      (default 4,512,640)
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rgbd_amd  # noqa: E402
from rgbd_amd import synth  # noqa: E402
from rgbd_amd._lib import lib  # noqa: E402

B, H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4,512,640").split(","))
L = lib()
net = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
net.load_state_dict(synth.synthetic_state_dict(0))
net.update(force=True)
net = net.to("cuda")
net.per_image_streams = True
r, d = synth.synthetic_batch(B, H, W, config_id=2)
rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
out = net.compress(rgb, depth)
L.rgbd_debug_conv_log(1)
out = net.compress(rgb, depth)
net.decompress(out["r_strings"], out["d_strings"], out["shape"])
L.rgbd_debug_conv_log(0)
n = L.rgbd_debug_conv_log_read(None, 0)
buf = ctypes.create_string_buffer(n)
L.rgbd_debug_conv_log_read(buf, n)
rows = [tuple(int(v) for v in ln.split(",")) for ln in buf.value.decode().strip().split("\n")[1:]]


def bench(key, N, iters=6):
    _, Hh, Ww, cin, cout, ntaps, stride, nphase, splitk = key
    k = int(round(ntaps ** 0.5))
    ms = ctypes.c_float(0)
    L.rgbd_debug_force_ckbd(nphase // 10)
    L.rgbd_debug_force_splitk(splitk)
    rc = L.rgbd_conv_bench(N, cin, Hh, Ww, cout, k, stride, k // 2, 1 if nphase % 10 > 1 else 0, 0, iters, ctypes.byref(ms))
    return ms.value if rc == 0 else float("inf")


t1 = t2 = 0.0
print("N,H,W,cin,cout,taps,stride,nphase,splitk  count   t(N) us   t(2N)/2 us  ratio")
for row in sorted(rows, key=lambda r: -r[9]):
    key, cnt = row[:9], row[9]
    a = min(bench(key, key[0]), bench(key, key[0]))
    b = min(bench(key, 2 * key[0]), bench(key, 2 * key[0])) / 2
    t1 += a * cnt
    t2 += b * cnt
    print(f"{key} x{cnt:3d} {a*1e3:9.1f} {b*1e3:9.1f}  {b/a:5.2f}", flush=True)
print(f"all conv launches of one enc+dec: separate {t1:.2f} ms, as double-batch launches {t2:.2f} ms ({t2/t1:.3f})")
