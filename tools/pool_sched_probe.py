#!/usr/bin/env python3
"""Scheduling probe for CodecPool.roundtrip_many: K steps of the c3 batch on W engine instances with a start stagger and a
cap on concurrent compress() calls.  Usage: pool_sched_probe.py K "W:stagger_ms:slots" ...   (prints ms/step per setting)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")  # one hardware queue per engine instance, as bench.py sets it
import torch  # noqa: E402

import rgbd_amd  # noqa: E402
from rgbd_amd import synth  # noqa: E402
from rgbd_amd.pool import CodecPool  # noqa: E402

K = int(sys.argv[1])
settings = [tuple(float(v) for v in a.split(":")) for a in sys.argv[2:]]
B, H, W_ = 4, 512, 640
sd = synth.synthetic_state_dict(0)
r, d = synth.synthetic_batch(B, H, W_, config_id=2)
rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
wmax = int(max(s[0] for s in settings))
pool = CodecPool(sd, config=rgbd_amd.model_config(), workers=wmax)
pool.roundtrip_many([(rgb, depth)] * wmax)
pool.roundtrip_many([(rgb, depth)] * wmax)
free, total = torch.cuda.mem_get_info()
print(f"instances {wmax}: HBM used {(total - free) / 2**30:.1f} GiB", flush=True)
for rep in range(2):
    for w, stg, slots in settings:
        full, pool.nets, pool.streams = (pool.nets, pool.streams), pool.nets[: int(w)], pool.streams[: int(w)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pool.roundtrip_many([(rgb, depth)] * K, stagger_ms=stg, compress_slots=int(slots))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pool.nets, pool.streams = full
        print(f"K={K} W={int(w)} stagger={stg} slots={int(slots)}: {dt * 1e3 / K:.2f} ms/step  {K * B * H * W_ / dt / 1e6:.2f} Mpx/s (padded px)", flush=True)
