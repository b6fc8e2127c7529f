#!/usr/bin/env python3
"""Index / symbol statistics of a synthetic recipe on one 512x640 pair: which scale-table rows the coder works on.
    python tools/recipe_probe.py [recipe] [gy gz gh gw gb]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import rgbd_amd  # noqa: E402
from rgbd_amd import synth  # noqa: E402

recipe = sys.argv[1] if len(sys.argv) > 1 else "high_rate"
if len(sys.argv) > 2:
    synth.HIGH_RATE_GAINS = tuple(float(v) for v in sys.argv[2:7])
net = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
net.load_state_dict(synth.synthetic_state_dict(0, recipe=recipe))
net.update(force=True)
net = net.to("cuda")
r, d = synth.synthetic_batch(1, 512, 640, config_id=3)
out = net.compress(torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda())
sizes = np.asarray(net.rgb_gaussian_conditional._cdf_length)
offs = np.asarray(net.rgb_gaussian_conditional._offset)
for m, nm in ((0, "rgb"), (1, "depth")):
    sym, idx = net.debug_symbols(m)
    v = sym - offs[idx]
    esc = (v < 0) | (v >= sizes[idx] - 2)
    h = np.bincount(idx, minlength=64)
    q = np.percentile(idx, [1, 25, 50, 75, 99])
    print(f"{recipe} {nm}: {len(sym)} symbols, index percentiles 1/25/50/75/99 = {q}, max {idx.max()}, share on rows > 128 slots "
          f"{(sizes[idx] - 1 > 128).mean():.3f}, escapes {esc.mean():.3f}, bytes {sum(len(s) for s in out['r_strings' if m == 0 else 'd_strings'][0])}")
    print("   histogram by 8:", [int(h[i:i + 8].sum()) for i in range(0, 64, 8)])
rec = net.decompress(out["r_strings"], out["d_strings"], out["shape"])
print("decoded ok", tuple(rec["x_hat"]["r"].shape))
