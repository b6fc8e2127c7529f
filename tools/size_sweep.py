#!/usr/bin/env python3
"""Robustness sweep over image sizes / batch sizes: compress -> decompress must run, be deterministic, and the decoder's
x_hat must equal the eval-mode forward() reconstruction (same kernels, no coder) wherever forward() is available."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rgbd_amd  # noqa: E402
from rgbd_amd import ELIC_united, synth  # noqa: E402

net = ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
net.load_state_dict(synth.synthetic_state_dict(0))
net.update(force=True)
net = net.to("cuda")
net.per_image_streams = True
bad = 0
for B, H, W in [(1, 128, 128), (3, 128, 192), (2, 192, 320), (1, 320, 448), (5, 256, 256), (1, 704, 1024), (1, 1024, 1536),
                (2, 512, 640), (16, 128, 128)]:
    r, d = synth.synthetic_batch(B, H, W, config_id=40 + B, smooth=(H > 300))
    rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
    o1 = net.compress(rgb, depth)
    o2 = net.compress(rgb, depth)
    rec = net.decompress(o1["r_strings"], o1["d_strings"], o1["shape"])
    same = o1["r_strings"] == o2["r_strings"] and o1["d_strings"] == o2["d_strings"]
    xr, xd = rec["x_hat"]["r"], rec["x_hat"]["d"]
    fin = bool(torch.isfinite(xr).all() and torch.isfinite(xd).all())
    fwd_ok = "n/a"
    try:
        with torch.no_grad():
            f = net(rgb, depth)
        fwd_ok = bool(torch.equal(f["x_hat"]["r"].clamp(0, 1), xr) and torch.equal(f["x_hat"]["d"].clamp(0, 1), xd))
    except Exception as e:  # noqa: BLE001
        fwd_ok = f"forward failed: {type(e).__name__}"
    nbytes = sum(len(s) for lst in o1["r_strings"] + o1["d_strings"] for s in lst)
    ok = same and fin and fwd_ok in (True, "n/a")
    bad += 0 if ok else 1
    print(f"B={B} {H}x{W}: deterministic={same} finite={fin} decode==forward: {fwd_ok}  bytes={nbytes} ({nbytes*8/(B*H*W):.2f} bpp)  {'OK' if ok else 'FAIL'}",
          flush=True)
print("SWEEP OK" if not bad else f"SWEEP FAILED ({bad})")
sys.exit(1 if bad else 0)
