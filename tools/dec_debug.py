#!/usr/bin/env python3
"""Small decodes with growing sizes (progress printed before each call): localises a decoder bug without long runs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgbd_amd  # noqa: E402,F401
from rgbd_amd import ans  # noqa: E402
from rgbd_amd.entropy_models import GaussianConditional, get_scale_table  # noqa: E402

gc = GaussianConditional()
gc.update_scale_table(get_scale_table(), force=True)
cdf, sizes, offsets = gc.numpy_tables()
t = ans.Tables(cdf, sizes, offsets)
rng = np.random.default_rng(0)
scales = np.exp(np.linspace(np.log(0.11), np.log(256), 64))
for name, lo, hi, gain in (("narrow", 0, 8, 1.0), ("escapes", 0, 30, 3.0), ("wide", 40, 64, 1.0), ("all", 0, 64, 2.0)):
    for N in (1, 2, 3, 63, 64, 65, 130, 1000, 20000):
        idx = rng.integers(lo, hi, N).astype(np.int32)
        sym = np.rint(rng.normal(0.0, 1.0, N) * scales[idx] * gain).astype(np.int32)
        s = ans._encode(t, sym, idx)
        print(name, N, "decoding", flush=True)
        d = ans.RansDecoder()
        d.set_stream(s)
        out = np.asarray(d.decode_stream(idx, cdf, sizes, offsets), dtype=np.int32)
        bad = np.nonzero(out != sym)[0]
        print("   ", "ok" if not len(bad) else f"MISMATCH first at {bad[0]}: got {out[bad[0]]} want {sym[bad[0]]} idx {idx[bad[0]]} ({len(bad)} bad)", flush=True)
