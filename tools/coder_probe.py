#!/usr/bin/env python3
"""rANS kernel probe: encodes / decodes N symbols per scale class through the stand-alone coder ABI so that
`rocprofv3 --kernel-trace` shows the per-symbol cost of the serial chain for narrow and wide CDF rows.
Usage: rocprofv3 --kernel-trace --output-format csv -d out -o p -- python3 tools/coder_probe.py [N]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgbd_amd  # noqa: E402,F401
from rgbd_amd import ans  # noqa: E402
from rgbd_amd.entropy_models import GaussianConditional, get_scale_table  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
gc = GaussianConditional()
gc.update_scale_table(get_scale_table(), force=True)
cdf, sizes, offsets = gc.numpy_tables()
t = ans.Tables(cdf, sizes, offsets)
rng = np.random.default_rng(0)
scales = np.exp(np.linspace(np.log(0.11), np.log(256), 64))
for name, lo, hi, gain in (("idx0-7", 0, 8, 1.0), ("idx16-23", 16, 24, 1.0), ("idx32-39", 32, 40, 1.0), ("idx48-55", 48, 56, 1.0),
                           ("idx0-63", 0, 64, 1.0), ("stress-like", 0, 40, 3.0), ("zeros", 0, 4, 0.0)):
    idx = rng.integers(lo, hi, N).astype(np.int32)
    sym = np.rint(rng.normal(0.0, 1.0, N) * scales[idx] * gain).astype(np.int32)
    s = ans._encode(t, sym, idx)
    d = ans.RansDecoder()
    d.set_stream(s)
    out = np.asarray(d.decode_stream(idx, cdf, sizes, offsets), dtype=np.int32)
    assert np.array_equal(out, sym), name
    print(name, "rows", sizes[lo:hi].min(), "-", sizes[lo:hi].max(), "bytes/sym", len(s) / N, flush=True)
