"""ctypes binding of oracle/librgbd_oracle.so (the plain-C restatement of the reference coder).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/rans_oracle.c.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _DecState(ctypes.Structure):
    _fields_ = [("x", ctypes.c_uint64), ("pos", ctypes.c_int64)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "librgbd_oracle.so")
    src = os.path.join(_HERE, "rans_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "librgbd_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        i32p = ctypes.POINTER(ctypes.c_int32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.orc_rans_encode.restype = ctypes.c_int64
        L.orc_rans_encode.argtypes = [i32p, i32p, ctypes.c_int64, i32p, ctypes.c_int32, i32p, i32p, u8p,
                                      ctypes.c_int64]
        L.orc_rans_count_items.restype = ctypes.c_int64
        L.orc_rans_count_items.argtypes = [i32p, i32p, ctypes.c_int64, i32p, ctypes.c_int32, i32p, i32p]
        L.orc_rans_dec_init.restype = ctypes.c_int
        L.orc_rans_dec_init.argtypes = [u8p, ctypes.c_int64, ctypes.POINTER(_DecState)]
        L.orc_rans_decode.restype = ctypes.c_int
        L.orc_rans_decode.argtypes = [u8p, ctypes.c_int64, ctypes.POINTER(_DecState), i32p, ctypes.c_int64,
                                      i32p, ctypes.c_int32, i32p, i32p, i32p]
        L.orc_pmf_to_quantized_cdf.restype = ctypes.c_int
        L.orc_pmf_to_quantized_cdf.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int32, ctypes.c_int32,
                                               ctypes.POINTER(ctypes.c_uint32)]
        _LIB = L
    return _LIB


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


class Tables:
    """A CDF table set: cdf int32 [n_rows, stride], sizes int32 [n_rows], offsets int32 [n_rows]."""

    def __init__(self, cdf, sizes, offsets):
        self.cdf = _i32(cdf)
        assert self.cdf.ndim == 2
        self.sizes = _i32(sizes).reshape(-1)
        self.offsets = _i32(offsets).reshape(-1)
        assert self.sizes.shape[0] == self.cdf.shape[0] == self.offsets.shape[0]

    @property
    def stride(self):
        return int(self.cdf.shape[1])


def rans_encode(symbols, indexes, t: Tables) -> bytes:
    sym, idx = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
    assert sym.shape == idx.shape
    n = int(sym.shape[0])
    L = lib()
    args = (_p(sym, ctypes.c_int32), _p(idx, ctypes.c_int32), n, _p(t.cdf, ctypes.c_int32), t.stride,
            _p(t.sizes, ctypes.c_int32), _p(t.offsets, ctypes.c_int32))
    items = L.orc_rans_count_items(*args)
    if items < 0:
        raise MemoryError("oracle encoder")
    cap = 4 * (items + 2)
    out = np.empty(cap, dtype=np.uint8)
    nb = L.orc_rans_encode(*args, _p(out, ctypes.c_uint8), cap)
    if nb < 0:
        raise RuntimeError(f"oracle encoder failed: {nb}")
    return out[:nb].tobytes()


class RansDecoder:
    """Stateful decoder: set_stream() then any number of decode_stream() calls (rans_interface.cpp:278-351)."""

    def __init__(self):
        self._buf = None
        self._st = _DecState()

    def set_stream(self, data: bytes):
        self._buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        rc = lib().orc_rans_dec_init(_p(self._buf, ctypes.c_uint8), int(self._buf.shape[0]), ctypes.byref(self._st))
        if rc:
            raise ValueError("stream shorter than 8 bytes")

    def decode_stream(self, indexes, t: Tables) -> np.ndarray:
        idx = _i32(indexes).reshape(-1)
        out = np.empty(idx.shape[0], dtype=np.int32)
        rc = lib().orc_rans_decode(_p(self._buf, ctypes.c_uint8), int(self._buf.shape[0]), ctypes.byref(self._st),
                                   _p(idx, ctypes.c_int32), int(idx.shape[0]), _p(t.cdf, ctypes.c_int32), t.stride,
                                   _p(t.sizes, ctypes.c_int32), _p(t.offsets, ctypes.c_int32),
                                   _p(out, ctypes.c_int32))
        if rc:
            raise RuntimeError("oracle decoder failed")
        return out

    @property
    def words_consumed(self) -> int:
        return int(self._st.pos)


def rans_decode(data: bytes, indexes, t: Tables) -> np.ndarray:
    d = RansDecoder()
    d.set_stream(data)
    return d.decode_stream(indexes, t)


def pmf_to_quantized_cdf(pmf, precision: int = 16) -> np.ndarray:
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32)).reshape(-1)
    out = np.zeros(p.shape[0] + 1, dtype=np.uint32)
    rc = lib().orc_pmf_to_quantized_cdf(_p(p, ctypes.c_float), int(p.shape[0]), int(precision),
                                        _p(out, ctypes.c_uint32))
    if rc:
        raise ValueError(f"pmf_to_quantized_cdf failed: {rc}")
    return out


def load_reference_coder():
    """The reference's own C++ coder built into oracle/_ref (None if that build is absent)."""
    import importlib.util

    d = os.path.join(_HERE, "_ref")
    mods = {}
    if not os.path.isdir(d):
        return None
    for stem in ("ans", "_CXX"):
        hit = [f for f in os.listdir(d) if f.startswith(stem + ".") and f.endswith(".so")]
        if not hit:
            return None
        spec = importlib.util.spec_from_file_location(stem, os.path.join(d, hit[0]))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[stem] = m
    return mods
