"""ctypes binding of oracle/libcpu_arith.so -- the C restatement of the third-party CPU arithmetic (oneDNN / MKL / Sleef
accumulation orders) the reference's float path ends in.  See the header of oracle/cpu_arith.c.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libcpu_arith.so")
    src = os.path.join(_HERE, "cpu_arith.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libcpu_arith.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_expf_u10.restype = ctypes.c_float
        _LIB.orc_expf_u10.argtypes = [ctypes.c_float]
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def direct_blocks(cin: int):
    """oneDNN's multi-tap kernels: one block per 16 input channels."""
    return [16] * (cin // 16) + ([cin % 16] if cin % 16 else [])


def conv2d(x, w, b, stride=1, pad=0, blocks=None, bias_mode=None, order=0):
    """x [N,C,H,W], w [O,C,KH,KW].  blocks: channels per block (default: 16 each); bias_mode: 0 sum-then-bias,
    1 (S_0 + bias) + S_1 ..., 2 first chain starts from the bias (defaults: 1 for kernels > 1, 2 for 1x1)."""
    x, w, b = _f32(x), _f32(w), _f32(b)
    N, C, H, W = x.shape
    O, _, KH, KW = w.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    if blocks is None:
        blocks = direct_blocks(C)
    if bias_mode is None:
        bias_mode = 2 if KH == 1 and KW == 1 else 1
    bnd = np.concatenate([[0], np.cumsum(blocks)]).astype(np.int32)
    assert bnd[-1] == C
    y = np.empty((N, O, OH, OW), np.float32)
    lib().orc_conv_blocks(_p(x), N, C, H, W, _p(w), O, KH, KW, _p(b), stride, pad, _p(bnd), len(blocks), order, bias_mode,
                          _p(y), OH, OW)
    return y


def conv2d_im2col(x, w, b, stride=1, pad=0, kblocks=None):
    """ATen's small-tensor path (im2col + sgemm): k = c*KH*KW + ky*KW + kx, cut into `kblocks` (lengths in k)."""
    x, w, b = _f32(x), _f32(w), _f32(b)
    N, C, H, W = x.shape
    O, _, KH, KW = w.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    K = C * KH * KW
    kblocks = [K] if kblocks is None else kblocks
    kb = np.concatenate([[0], np.cumsum(kblocks)]).astype(np.int32)
    assert kb[-1] == K
    y = np.empty((N, O, OH, OW), np.float32)
    lib().orc_conv_im2col_kblocks(_p(x), N, C, H, W, _p(w), O, KH, KW, _p(b), stride, pad, _p(kb), len(kblocks), _p(y), OH, OW)
    return y


def deconv2d_s1(x, w, b, pad):
    """conv_transpose2d, stride 1; w [C,O,KH,KW]."""
    x, w, b = _f32(x), _f32(w), _f32(b)
    N, C, H, W = x.shape
    _, O, KH, KW = w.shape
    y = np.empty((N, O, H - 1 - 2 * pad + KH, W - 1 - 2 * pad + KW), np.float32)
    lib().orc_deconv_s1(_p(x), N, C, H, W, _p(w), O, KH, KW, _p(b), pad, _p(y))
    return y


def sigmoid(x, threads=None):
    """torch.sigmoid of a contiguous tensor.  threads=None: the vector body everywhere (what a tensor without scalar tails gets);
    threads=T: as T CPU threads compute it, scalar tails of the parallel chunks included (orc_sigmoid_tensor)."""
    x = _f32(x)
    y = np.empty_like(x)
    if threads is None:
        lib().orc_sigmoid(_p(x), _p(y), ctypes.c_int64(x.size))
    else:
        lib().orc_sigmoid_tensor(_p(x), _p(y), ctypes.c_int64(x.size), ctypes.c_int(threads))
    return y
