"""Reference-arithmetic tables: the shape-dependent part of the accumulation order of the CPU kernels the
reference runs on, measured on the reference machine by tools/refarith/discover.py and committed as data
(refarith_tables.json).  This module only hands the entries to the engine (rgbd_elic_set_ref_blocks)."""
import ctypes
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_TABLES = None


def tables():
    global _TABLES
    if _TABLES is None:
        with open(os.path.join(_HERE, "refarith_tables.json")) as f:
            _TABLES = json.load(f)
    return _TABLES


def push(L, handle, check):
    """L: the loaded library; handle: rgbd_elic*; every table entry goes to the engine (kind 0 = 1x1 reduce blocks)."""
    if L.rgbd_elic_get_refnum(handle) != 1:
        return 0
    n = 0
    # kind 4: the thread count of the reference run the tables were measured under (torch.sigmoid's chunking depends on it)
    thr = (ctypes.c_int32 * 1)(int(tables()["meta"].get("threads", 8)))
    check(L.rgbd_elic_set_ref_blocks(handle, 4, 0, 0, 0, 0, 0, thr, 1), "set_ref_blocks")
    for cin, cout, h, w, b, blocks in tables()["conv1x1"]:
        arr = (ctypes.c_int32 * len(blocks))(*blocks)
        check(L.rgbd_elic_set_ref_blocks(handle, 0, cin, cout, h, w, b, arr, len(blocks)), "set_ref_blocks")
        n += 1
    # kind 1: K blocks of the small-tensor route (im2col + sgemm), keyed by (cin, cout, h, w, k * 100 + stride * 10 + pad)
    for cin, cout, k, h, w, stride, pad, blocks in tables()["im2col"]:
        if len(blocks) > 16:
            continue
        arr = (ctypes.c_int32 * len(blocks))(*blocks)
        check(L.rgbd_elic_set_ref_blocks(handle, 1, cin, cout, h, w, k * 100 + stride * 10 + pad, arr, len(blocks)), "set_ref_blocks")
        n += 1
    # kind 2: SE_Block Linear layers on one vector: dot-product class per output row, run-length encoded
    for K, J, rle in tables().get("linear", []):
        cls = []
        for c, cnt in zip(rle[0::2], rle[1::2]):
            cls += [c] * cnt
        arr = (ctypes.c_int32 * len(cls))(*cls)
        check(L.rgbd_elic_set_ref_blocks(handle, 2, K, J, 0, 0, 1, arr, len(cls)), "set_ref_blocks")
        n += 1
    # kind 3: stride-2 transposed convs of the hyper-synthesis: tap chains per (phase, column), flattened
    for cin, cout, k, h, w, b, flat in tables().get("deconv_s2", []):
        arr = (ctypes.c_int32 * len(flat))(*flat)
        check(L.rgbd_elic_set_ref_blocks(handle, 3, cin, cout, h, w, b, arr, len(flat)), "set_ref_blocks")
        n += 1
    return n
