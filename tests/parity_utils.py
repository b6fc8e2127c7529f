"""Box-independent parity bookkeeping against the REFERENCE's golden streams (tests/golden/*.npz).

The oracle's float path runs on whatever CPU the GPU box has (oneDNN picks kernels by ISA), so "parts identical to the
oracle" can differ from box to box.  The golden y-streams were produced once by the unmodified reference; they are pure
data.  `golden_parts_identical` decodes such a stream part by part with the oracle's integer decoder, feeding it the
GPU's own indexes, and counts how many parts (in coding order: rgb anchor, depth anchor, rgb non-anchor, depth non-anchor
per slice -- models/elic_united.py:265-348) reproduce the GPU's symbols exactly before the first difference.  A count
of all parts together with equal stream lengths means the GPU's stream IS the reference's stream.
"""
import json
import os

import numpy as np

from oracle import coder

FLOORS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_floors.json")
CONTRACT_DPSNR = 1e-4  # dB; BASELINE.json north_star: "within 1e-4 PSNR for the float reconstruction"


def floors():
    with open(FLOORS_PATH) as f:
        return json.load(f)


def part_sizes(slice_ch, h, w, B=1):
    """Symbols per part and modality for a call whose y-stream covers B images (B=1 for per-image streams)."""
    out = []
    for C in slice_ch:
        n = B * C * h * (w // 2)
        out += [n, n]  # anchor, non-anchor
    return out


def golden_parts_identical(gsym, gidx, golden, tables, sizes, modalities=2):
    """gsym / gidx: {mod: int32 array of one stream's symbols / indexes in stream order}; golden: {mod: bytes}.
    Returns (#parts identical in coding order before the first difference, total parts)."""
    dec, pos = {}, {}
    for m in range(modalities):
        dec[m] = coder.RansDecoder()
        dec[m].set_stream(golden[m])
        pos[m] = 0
    clean, total = 0, len(sizes) * modalities
    for k, n in enumerate(sizes):  # k = 2 * slice + (0 anchor | 1 non-anchor)
        for m in range(modalities):
            a, b = pos[m], pos[m] + n
            pos[m] = b
            try:
                got = dec[m].decode_stream(gidx[m][a:b], tables)
            except RuntimeError:  # ran off the stream: the indexes already differ from the reference's
                return clean, total
            if not np.array_equal(got, gsym[m][a:b]):
                return clean, total
            clean += 1
    return clean, total
