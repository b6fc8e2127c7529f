"""`compressai.ans` / `compressai._CXX` look-alikes on top of the C ABI (GPU coder, no CPU fallback).

Same class names, method names, list-based arguments and return types as the reference's pybind11 modules
(CompressAI/compressai/cpp_exts/rans/rans_interface.cpp:353-373, cpp_exts/ops/ops.cpp:83-90), so the reference's
own Python glue can run against this coder unchanged.
"""
import ctypes

import numpy as np

from ._lib import check, lib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _ptr(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def pmf_to_quantized_cdf(pmf, precision: int = 16):
    """List[float] -> List[int] (len+1 entries), as compressai._CXX.pmf_to_quantized_cdf."""
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32)).reshape(-1)
    out = np.zeros(p.shape[0] + 1, dtype=np.uint32)
    check(lib().rgbd_pmf_to_quantized_cdf(_ptr(p, ctypes.c_float), int(p.shape[0]), int(precision),
                                          _ptr(out, ctypes.c_uint32)), "pmf_to_quantized_cdf")
    return out.tolist()


class Tables:
    """CDF rows packed and resident on the GPU (built once per table set, cached by content)."""

    _cache = {}

    def __init__(self, cdfs, cdfs_sizes, offsets):
        sizes = _i32(cdfs_sizes).reshape(-1)
        offs = _i32(offsets).reshape(-1)
        if isinstance(cdfs, np.ndarray):
            cdf = _i32(cdfs)
        else:
            stride = max(len(r) for r in cdfs)
            cdf = np.zeros((len(cdfs), stride), dtype=np.int32)
            for i, r in enumerate(cdfs):
                cdf[i, : len(r)] = r
        self.n_rows = int(cdf.shape[0])
        self._h = ctypes.c_void_p()
        check(lib().rgbd_tables_create(_ptr(cdf, ctypes.c_int32), int(cdf.shape[1]), _ptr(sizes, ctypes.c_int32),
                                       _ptr(offs, ctypes.c_int32), self.n_rows, ctypes.byref(self._h)), "tables_create")

    @classmethod
    def cached(cls, cdfs, cdfs_sizes, offsets):
        arr = cdfs if isinstance(cdfs, np.ndarray) else None
        key = (id(cdfs), len(cdfs), hash(bytes(_i32(cdfs_sizes).tobytes())), hash(bytes(_i32(offsets).tobytes())),
               hash(arr.tobytes()) if arr is not None else hash(tuple(cdfs[-1])))
        t = cls._cache.get(key)
        if t is None:
            if len(cls._cache) > 16:
                cls._cache.clear()
            t = cls._cache[key] = cls(cdfs, cdfs_sizes, offsets)
        return t

    @property
    def handle(self):
        return self._h

    def __del__(self):
        try:
            if self._h:
                lib().rgbd_tables_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _encode(t: Tables, symbols, indexes) -> bytes:
    sym, idx = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
    if sym.shape != idx.shape:
        raise ValueError("symbols and indexes must have the same length")
    n = int(sym.shape[0])
    cap = int(lib().rgbd_rans_max_bytes(n))
    out = np.empty(cap, dtype=np.uint8)
    ln = ctypes.c_int64(0)
    check(lib().rgbd_rans_encode(t.handle, _ptr(sym, ctypes.c_int32), _ptr(idx, ctypes.c_int32), n,
                                 _ptr(out, ctypes.c_uint8), cap, ctypes.byref(ln)), "rans_encode")
    return out[: ln.value].tobytes()


class BufferedRansEncoder:
    """rans_interface.hpp:34-53: encode_with_indexes() appends, flush() emits one stream and clears."""

    def __init__(self):
        self._sym, self._idx, self._tables = [], [], None

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        self._sym.append(_i32(symbols).reshape(-1))
        self._idx.append(_i32(indexes).reshape(-1))
        self._tables = Tables.cached(cdfs, cdfs_sizes, offsets)

    def flush(self) -> bytes:
        if self._tables is None:
            raise ValueError("flush() before encode_with_indexes()")
        s = _encode(self._tables, np.concatenate(self._sym), np.concatenate(self._idx))
        self._sym, self._idx = [], []
        return s


class RansEncoder:
    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets) -> bytes:
        return _encode(Tables.cached(cdfs, cdfs_sizes, offsets), symbols, indexes)


class RansDecoder:
    def __init__(self):
        self._h = ctypes.c_void_p()
        check(lib().rgbd_rans_decoder_create(ctypes.byref(self._h)), "decoder_create")

    def set_stream(self, encoded: bytes):
        buf = np.frombuffer(bytes(encoded), dtype=np.uint8)
        check(lib().rgbd_rans_decoder_set_stream(self._h, _ptr(buf, ctypes.c_uint8), int(buf.shape[0])), "set_stream")

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets):
        t = Tables.cached(cdfs, cdfs_sizes, offsets)
        idx = _i32(indexes).reshape(-1)
        out = np.empty(idx.shape[0], dtype=np.int32)
        check(lib().rgbd_rans_decoder_decode(self._h, t.handle, _ptr(idx, ctypes.c_int32), int(idx.shape[0]),
                                             _ptr(out, ctypes.c_int32)), "decode_stream")
        return out.tolist()

    def decode_with_indexes(self, encoded, indexes, cdfs, cdfs_sizes, offsets):
        self.set_stream(encoded)
        return self.decode_stream(indexes, cdfs, cdfs_sizes, offsets)

    def __del__(self):
        try:
            if self._h:
                lib().rgbd_rans_decoder_destroy(self._h)
                self._h = None
        except Exception:
            pass
