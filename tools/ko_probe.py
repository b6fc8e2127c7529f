#!/usr/bin/env python3
"""Timing-only decode of one narrow-row class (results are NOT checked: used with knock-out builds of the decoder loop,
RGBD_AMD_LIB=<variant>.so, to price single instructions of the hot loop).  Prints ns/symbol from hipEvent-free wall time of the
stand-alone decode call minus nothing: read it from a rocprofv3 kernel trace, or use the printed wall figure for a rough view."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgbd_amd  # noqa: E402,F401
from rgbd_amd import ans  # noqa: E402
from rgbd_amd.entropy_models import GaussianConditional, get_scale_table  # noqa: E402

N = 400_000
gc = GaussianConditional()
gc.update_scale_table(get_scale_table(), force=True)
cdf, sizes, offsets = gc.numpy_tables()
rng = np.random.default_rng(0)
scales = np.exp(np.linspace(np.log(0.11), np.log(256), 64))
idx = rng.integers(16, 24, N).astype(np.int32)
sym = np.rint(rng.normal(0.0, 1.0, N) * scales[idx]).astype(np.int32)
s = open(sys.argv[1], "rb").read() if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else None
if s is None:
    t = ans.Tables(cdf, sizes, offsets)
    s = ans._encode(t, sym, idx)
    if len(sys.argv) > 1:
        open(sys.argv[1], "wb").write(s)
best = 1e9
for _ in range(3):
    d = ans.RansDecoder()
    d.set_stream(s)
    t0 = time.perf_counter()
    out = d.decode_stream(idx, cdf, sizes, offsets)
    best = min(best, time.perf_counter() - t0)
print(os.environ.get("RGBD_AMD_LIB", "default")[-12:], f"{best / N * 1e9:7.1f} ns/symbol wall (incl. copies)", flush=True)
