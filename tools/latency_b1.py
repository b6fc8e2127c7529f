#!/usr/bin/env python3
"""Single-image latency (B=1, one engine instance): the reference tester's calling pattern (tester_united.py:141-195)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rgbd_amd  # noqa: E402
from rgbd_amd import ELIC_united, synth  # noqa: E402

H, W = (int(v) for v in sys.argv[1:3]) if len(sys.argv) >= 3 else (256, 256)
net = ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
net.load_state_dict(synth.synthetic_state_dict(0))
net.update(force=True)
net = net.to("cuda")
r, d = synth.synthetic_batch(1, H, W, config_id=2)
rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
for _ in range(3):
    out = net.compress(rgb, depth)
    net.decompress(out["r_strings"], out["d_strings"], out["shape"])
N = 10
torch.cuda.synchronize()
te = td = 0.0
for _ in range(N):
    t0 = time.perf_counter()
    out = net.compress(rgb, depth)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    net.decompress(out["r_strings"], out["d_strings"], out["shape"])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    te += t1 - t0
    td += t2 - t1
print(f"B=1 {H}x{W}: enc {te/N*1e3:.2f} ms  dec {td/N*1e3:.2f} ms  -> {H*W/((te+td)/N)/1e6:.2f} Mpx/s; y bytes {len(out['r_strings'][0][0])}")
