#!/usr/bin/env python3
"""Per-layer conv timing inside a real compress()+decompress() (HIP events around every launch, single instance)."""
import ctypes
import os
import re
import sys
import collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rgbd_amd  # noqa: E402
from rgbd_amd import ELIC_united, synth  # noqa: E402
from rgbd_amd._lib import _SO  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 256, 256)
sd = synth.synthetic_state_dict(0)
net = ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
net.load_state_dict(sd)
net.update(force=True)
net = net.to("cuda")
if os.environ.get("LAYER_FORCE_FUSE"):  # rgbd_debug_force_fuse mode for the whole run (A/B of the fused-tail kernels)
    from rgbd_amd._lib import lib

    lib().rgbd_debug_force_fuse(int(os.environ["LAYER_FORCE_FUSE"]))
net.per_image_streams = True
r, d = synth.synthetic_batch(B, H, W, config_id=2)
rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
for _ in range(2):
    out = net.compress(rgb, depth)
    net.decompress(out["r_strings"], out["d_strings"], out["shape"])
net.set_profile(True)
N = 3
for _ in range(N):
    out = net.compress(rgb, depth)
    net.decompress(out["r_strings"], out["d_strings"], out["shape"])
L = ctypes.CDLL(_SO)
L.rgbd_elic_profile_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
path = "/tmp/layers.csv"
L.rgbd_elic_profile_dump(net._h, path.encode())
rows = [l.strip().split(",") for l in open(path)][1:]
groups = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])


def group(name):
    name = re.sub(r"\.\d+\.", ".N.", name)
    name = re.sub(r"(rgb|depth)_", "M_", name)
    name = re.sub(r"\.(r|d)_", ".M_", name)
    return re.sub(r"\.\d+$", ".N", name)


for name, cnt, ms, gf, tf, gfx, tfx in rows:
    g = groups[group(name)]
    g[0] += int(cnt)
    g[1] += float(ms)
    g[2] += float(gf)
    g[3] += float(gfx)
tot = sum(g[1] for g in groups.values())
print(f"total conv ms/step {tot/N:.2f}   (TF/s: algorithmic = the reference's layer FLOPs / time; executed = what the launches compute / time --")
print("   lower where a launch computes one checkerboard half of its layer or meets only half of the taps with non-zero inputs)")
for k, g in sorted(groups.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:62s} n={g[0]//N:4d} {g[1]/N:7.3f} ms/step {g[2]/max(g[1],1e-9):7.1f} TF/s  executed {g[3]/max(g[1],1e-9):6.1f}  {100*g[1]/tot:5.1f}%")
if os.environ.get("LAYER_RAW"):  # LAYER_RAW=<regex>: the individual layers behind the groups
    pat = re.compile(os.environ["LAYER_RAW"])
    for name, cnt, ms, gf, tf, gfx, tfx in sorted(rows, key=lambda r: r[0]):
        if pat.search(name):
            print(f"  {name:70s} n={int(cnt)//N:3d} {float(ms)/N*1e3:8.1f} us/step {float(gf)/max(float(ms),1e-9):7.1f} TF/s  executed {float(gfx)/max(float(ms),1e-9):6.1f}")
