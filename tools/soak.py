#!/usr/bin/env python3
"""Soak check: many concurrent compress()/decompress() round trips through a CodecPool must all give the same streams and
the same reconstruction as a single engine instance (races between instances, coder fall-back paths, arena reuse).
Usage: soak.py [rounds] [workers]"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rgbd_amd  # noqa: E402
from rgbd_amd import CodecPool, synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sd = synth.synthetic_state_dict(0)
pool = CodecPool(sd, config=rgbd_amd.model_config(), workers=workers, per_image_streams=True)
batches = []
for i in range(4):  # four different inputs / sizes
    B, H, W = ((8, 256, 256), (4, 256, 320), (2, 128, 192), (3, 192, 256))[i]
    r, d = synth.synthetic_batch(B, H, W, config_id=20 + i)
    batches.append((torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()))
ref = []
for rgb, depth in batches:  # single instance, nothing else running
    out = pool.nets[0].compress(rgb, depth)
    rec = pool.nets[0].decompress(out["r_strings"], out["d_strings"], out["shape"])
    ref.append((out, rec["x_hat"]["r"].clone(), rec["x_hat"]["d"].clone()))
bad = 0
for rnd in range(rounds):
    res = pool.roundtrip_many([batches[k % 4] for k in range(2 * workers)])
    for k, (out, xr, xd) in enumerate(res):
        o, rr, rd = ref[k % 4]
        if out["r_strings"] != o["r_strings"] or out["d_strings"] != o["d_strings"] or not torch.equal(xr, rr) or not torch.equal(xd, rd):
            bad += 1
    print(f"round {rnd}: {2 * workers} round trips, mismatches so far {bad}", flush=True)
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
