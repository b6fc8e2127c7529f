"""The blocked-accumulation conv kernels (csrc/conv_mfma_blk.hip, through the C ABI rgbd_conv2d_ref_nchw) against the oracle's
C restatement of the reference's CPU arithmetic (oracle/cpu_arith.c) -- BIT FOR BIT (np.array_equal on the fp32 outputs).

oracle/cpu_arith.c itself is pinned to torch CPU (the library stack the reference runs on) in tests/test_oracle_arith.py
and, on frozen inputs, through tests/golden/refarith_pins.npz."""
import ctypes

import numpy as np
import pytest
import torch

from gpu_utils import require_gpu

pytestmark = pytest.mark.gpu

f32p = ctypes.POINTER(ctypes.c_float)
i32p = ctypes.POINTER(ctypes.c_int32)


def gpu_conv_ref(x, w, b, stride, pad, transposed=0, act=0, res=None, blocks=None, bias_mode=1, flags=0, dev=None):
    from rgbd_amd._lib import check, lib

    n, cin, h, wd = x.shape
    cout = w.shape[1] if transposed else w.shape[0]
    k = w.shape[-1]
    if transposed:
        oh, ow = (h - 1) * stride - 2 * pad + k + stride - 1, (wd - 1) * stride - 2 * pad + k + stride - 1
    else:
        oh, ow = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    xd = torch.from_numpy(x).to(dev).contiguous()
    yd = torch.empty((n, cout, oh, ow), device=dev)
    rd = torch.from_numpy(res).to(dev).contiguous() if res is not None else None
    wn = np.ascontiguousarray(w, np.float32)
    bn = np.ascontiguousarray(b, np.float32)
    bl = np.ascontiguousarray(blocks, np.int32) if blocks is not None else None
    check(lib().rgbd_conv2d_ref_nchw(ctypes.c_void_p(xd.data_ptr()), n, cin, h, wd, wn.ctypes.data_as(f32p),
                                     bn.ctypes.data_as(f32p), cout, k, stride, pad, transposed, act,
                                     ctypes.c_void_p(rd.data_ptr()) if rd is not None else None,
                                     ctypes.c_void_p(yd.data_ptr()), None,
                                     bl.ctypes.data_as(i32p) if bl is not None else None, 0 if bl is None else len(bl),
                                     bias_mode, flags), "conv2d_ref")
    return yd.cpu().numpy()


def _data(seed, n, cin, h, w, cout, k, transposed=False):
    rng = np.random.RandomState(seed)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    ws = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    wt = (rng.standard_normal(ws) / (cin * k * k) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.5).astype(np.float32)
    return x, wt, b


DIRECT = [
    # n, cin, h, w, cout, k, stride, pad   (oneDNN jit:avx512_core structure: a block per 16 channels, (S_0 + bias) + S_1 ...)
    (1, 96, 32, 40, 96, 3, 1, 1),
    (2, 192, 16, 24, 96, 3, 1, 1),
    (1, 384, 32, 48, 192, 5, 2, 2),
    (1, 48, 33, 47, 48, 3, 2, 0),
    (1, 213, 8, 12, 42, 3, 1, 1),
    (1, 42, 8, 12, 32, 5, 1, 2),
    (1, 512, 16, 16, 384, 5, 1, 2),
    (1, 320, 16, 20, 192, 3, 1, 1),
    (3, 16, 16, 16, 224, 5, 1, 2),
]


@pytest.mark.parametrize("case", DIRECT, ids=[str(c) for c in DIRECT])
def test_direct_conv_blocked_bit_exact(case):
    dev = require_gpu()
    from oracle import cpu_arith as ca

    n, cin, h, w, cout, k, stride, pad = case
    x, wt, b = _data(hash(case) % 1000, n, cin, h, w, cout, k)
    want = ca.conv2d(x, wt, b, stride, pad)
    got = gpu_conv_ref(x, wt, b, stride, pad, dev=dev)
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} outputs differ, max {np.abs(got - want).max()}"


ONE_BY_ONE = [
    # n, cin, h, w, cout, blocks  (jit_1x1:avx512_core: reduce blocks, the first chain starts from the bias)
    (1, 192, 32, 32, 96, [112, 80]),
    (2, 192, 64, 80, 96, [96, 96]),
    (1, 96, 64, 80, 192, [96]),
    (1, 1280, 32, 40, 213, [384, 384, 384, 128]),
    (1, 2816, 16, 16, 469, [448] * 6 + [128]),
    (1, 320, 16, 16, 160, [320]),
    (1, 48, 64, 80, 192, [48]),
]


@pytest.mark.parametrize("case", ONE_BY_ONE, ids=[str(c) for c in ONE_BY_ONE])
@pytest.mark.parametrize("split", [0, 1])
def test_one_by_one_blocked_bit_exact(case, split):
    dev = require_gpu()
    from oracle import cpu_arith as ca

    n, cin, h, w, cout, blocks = case
    x, wt, b = _data(7 + cin, n, cin, h, w, cout, 1)
    want = ca.conv2d(x, wt, b, 1, 0, blocks=blocks, bias_mode=2)
    got = gpu_conv_ref(x, wt, b, 1, 0, blocks=blocks, bias_mode=2, flags=2 if split else 0, dev=dev)
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} outputs differ, max {np.abs(got - want).max()}"


def test_deconv_stride1_blocked_bit_exact():
    dev = require_gpu()
    from oracle import cpu_arith as ca

    for (n, cin, h, w, cout) in [(1, 960, 8, 12, 640), (2, 64, 24, 24, 64)]:
        x, wt, b = _data(3, n, cin, h, w, cout, 3, transposed=True)
        want = ca.deconv2d_s1(x, wt, b, 1)
        got = gpu_conv_ref(x, wt, b, 1, 1, transposed=1, bias_mode=0, dev=dev)
        assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} outputs differ"


def test_sigmoid_gate_epilogue_bit_exact():
    """AttentionBlock's conv_b.3 (layers.py:198-213): a * sigmoid(conv1x1(b)) + x with the CPU's vector sigmoid."""
    dev = require_gpu()
    from oracle import cpu_arith as ca

    x, wt, b = _data(11, 1, 192, 32, 40, 192, 1)
    want = ca.sigmoid(ca.conv2d(x, wt, b, 1, 0, blocks=[192], bias_mode=2))
    got = gpu_conv_ref(x, wt, b, 1, 0, act=3, blocks=[192], bias_mode=2, flags=1, dev=dev)
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} outputs differ"


def test_tile_choice_does_not_change_a_bit_blocked():
    dev = require_gpu()
    from oracle import cpu_arith as ca
    from rgbd_amd._lib import RgbdError, lib

    ran = 0
    x, wt, b = _data(5, 2, 96, 32, 40, 96, 3)
    want = ca.conv2d(x, wt, b, 1, 1)
    try:
        for cfg in (b"2,2,4,16,1", b"2,4,2,16,0", b"1,3,4,16,1", b"2,1,1,16,0", b"2,1,1,64,0", b"1,1,2,16,1", b"2,5,2,16,1", b"1,1,1,16,1"):
            lib().rgbd_debug_force_tile(cfg)
            try:
                got = gpu_conv_ref(x, wt, b, 1, 1, dev=dev)
            except RgbdError as e:  # (a tile whose stage does not fit this layer: not a numerics matter)
                assert "-28" in str(e), e
                continue
            ran += 1
            assert np.array_equal(got, want), cfg
        assert ran >= 5
        x, wt, b = _data(6, 2, 192, 32, 32, 96, 1)
        want = ca.conv2d(x, wt, b, 1, 0, blocks=[112, 80], bias_mode=2)
        for cfg in (b"2,2,8,16,4", b"2,1,8,16,5", b"1,3,4,16,4", b"2,4,4,16,5", b"2,2,2,16,1", b"1,2,1,16,0", b"2,1,1,64,0", b"1,1,1,64,0",
                    b"2,1,2,64,0", b"1,1,1,16,4"):
            lib().rgbd_debug_force_tile(cfg)
            try:
                got = gpu_conv_ref(x, wt, b, 1, 0, blocks=[112, 80], bias_mode=2, dev=dev)
            except RgbdError as e:
                assert "-28" in str(e), e
                continue
            ran += 1
            assert np.array_equal(got, want), cfg
        assert ran >= 13
    finally:
        lib().rgbd_debug_force_tile(b"")
