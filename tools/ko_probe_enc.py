#!/usr/bin/env python3
"""Timing-only encode of one narrow-row class (the stream is NOT checked: used with knock-out builds of the encoder's item
walk, RGBD_AMD_LIB=<variant>.so).  Read ns/symbol of rans_encode_kernel* from a rocprofv3 kernel trace."""
import os
import sys

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
t = ans.Tables(cdf, sizes, offsets)
for _ in range(3):
    try:
        s = ans._encode(t, sym, idx)
    except Exception as e:  # a knock-out build may overflow its output buffer: the timing is still there
        print("encode raised", type(e).__name__)
print(os.environ.get("RGBD_AMD_LIB", "default")[-12:], len(s) if "s" in dir() else -1, flush=True)
