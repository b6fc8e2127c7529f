"""MFMA conv / transposed-conv kernel (via the C ABI) vs torch CPU fp32 convolution.  Tolerance: the kernel is an
exact-fp32 fma chain; against oneDNN's different summation order the error is ~1e-6 relative to sum|a.b|, so we allow
2e-5 * (max|ref| + 1e-3) absolute."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_utils import require_gpu

pytestmark = pytest.mark.gpu

CASES = [
    # n, cin, h, w, cout, k, stride, pad, transposed, act, residual
    (2, 3, 64, 64, 192, 5, 2, 2, 0, 0, 0),
    (1, 1, 64, 96, 192, 5, 2, 2, 0, 0, 0),
    (2, 96, 32, 48, 96, 3, 1, 1, 0, 1, 0),
    (1, 192, 32, 32, 96, 1, 1, 0, 0, 1, 0),
    (1, 96, 32, 32, 192, 1, 1, 0, 0, 0, 1),
    (1, 384, 32, 48, 192, 5, 2, 2, 0, 0, 0),
    (1, 48, 33, 47, 48, 3, 2, 0, 0, 0, 0),
    (2, 48, 9, 11, 48, 3, 1, 1, 0, 1, 0),
    (1, 320, 8, 12, 192, 5, 2, 2, 1, 0, 0),
    (1, 192, 16, 24, 3, 5, 2, 2, 1, 0, 0),
    (1, 192, 16, 16, 1, 5, 2, 2, 1, 0, 0),
    (1, 960, 8, 12, 640, 3, 1, 1, 1, 0, 0),
    (2, 384, 2, 3, 320, 5, 2, 2, 1, 2, 0),
    (1, 1344, 8, 12, 224, 1, 1, 0, 0, 1, 0),
    (1, 213, 8, 12, 42, 3, 1, 1, 0, 1, 0),
    (1, 42, 8, 12, 32, 5, 1, 2, 0, 0, 0),
    (1, 128, 16, 16, 224, 5, 1, 2, 0, 1, 0),
    (1, 320, 8, 40, 160, 1, 1, 0, 0, 3, 0),
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv_vs_torch(case):
    dev = require_gpu()
    from rgbd_amd._lib import check, lib

    n, cin, h, w, cout, k, stride, pad, tr, act, use_res = case
    g = torch.Generator().manual_seed(hash(case) % (2**31))
    x = torch.randn(n, cin, h, w, generator=g)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    wt = torch.randn(*wshape, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    if tr:
        ref = F.conv_transpose2d(x, wt, b, stride=stride, padding=pad, output_padding=stride - 1)
    else:
        ref = F.conv2d(x, wt, b, stride=stride, padding=pad)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + res
    ref = [lambda t: t, torch.relu, lambda t: F.leaky_relu(t, 0.01), torch.sigmoid][act](ref)
    xd = x.to(dev).contiguous()
    yd = torch.empty(ref.shape, device=dev)
    rd = res.to(dev).contiguous() if res is not None else None
    f32p = ctypes.POINTER(ctypes.c_float)
    wn, bn = wt.contiguous().numpy(), b.numpy()
    check(lib().rgbd_conv2d_nchw(ctypes.c_void_p(xd.data_ptr()), n, cin, h, w, wn.ctypes.data_as(f32p),
                                 bn.ctypes.data_as(f32p), cout, k, stride, pad, tr, act,
                                 ctypes.c_void_p(rd.data_ptr()) if rd is not None else None,
                                 ctypes.c_void_p(yd.data_ptr()), None), "conv2d")
    got = yd.cpu()
    tol = 2e-5 * (ref.abs().max().item() + 1e-3)
    err = (got - ref).abs().max().item()
    assert err <= tol, f"max abs err {err} > {tol}"


def test_conv_deterministic_and_batch_invariant():
    dev = require_gpu()
    from rgbd_amd._lib import check, lib

    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 96, 16, 24, generator=g)
    wt = (torch.randn(96, 96, 3, 3, generator=g) / 30).contiguous()
    b = torch.randn(96, generator=g)
    f32p = ctypes.POINTER(ctypes.c_float)

    def run(xx):
        xd = xx.to(dev).contiguous()
        yd = torch.empty((xx.shape[0], 96, 16, 24), device=dev)
        check(lib().rgbd_conv2d_nchw(ctypes.c_void_p(xd.data_ptr()), xx.shape[0], 96, 16, 24,
                                     wt.numpy().ctypes.data_as(f32p), b.numpy().ctypes.data_as(f32p), 96, 3, 1, 1, 0, 0,
                                     None, ctypes.c_void_p(yd.data_ptr()), None), "conv2d")
        return yd.cpu()

    full = run(x)
    assert torch.equal(full, run(x))
    assert torch.equal(full[1:2], run(x[1:2]))  # same bits whatever the batch the image rides in


@pytest.mark.parametrize("split", [2, 3, 8])
def test_split_k_matches_and_is_deterministic(split):
    """Split-K (used for the weight-heavy entropy-model layers): partial chains summed in a fixed order."""
    dev = require_gpu()
    from rgbd_amd._lib import check, lib

    g = torch.Generator().manual_seed(17)
    x = torch.randn(2, 224, 16, 16, generator=g)
    wt = (torch.randn(128, 224, 5, 5, generator=g) / 75).contiguous()
    b = torch.randn(128, generator=g)
    ref = torch.relu(F.conv2d(x, wt, b, padding=2))
    f32p = ctypes.POINTER(ctypes.c_float)

    def run(xx):
        xd = xx.to(dev).contiguous()
        yd = torch.empty((xx.shape[0], 128, 16, 16), device=dev)
        check(lib().rgbd_conv2d_nchw(ctypes.c_void_p(xd.data_ptr()), xx.shape[0], 224, 16, 16, wt.numpy().ctypes.data_as(f32p),
                                     b.numpy().ctypes.data_as(f32p), 128, 5, 1, 2, 0, 1, None, ctypes.c_void_p(yd.data_ptr()),
                                     None), "conv2d")
        return yd.cpu()

    lib().rgbd_debug_force_splitk(split)
    try:
        got = run(x)
        assert (got - ref).abs().max().item() <= 2e-5 * (ref.abs().max().item() + 1e-3)
        assert torch.equal(got, run(x)) and torch.equal(got[1:2], run(x[1:2]))
    finally:
        lib().rgbd_debug_force_splitk(0)


@pytest.mark.gpu
@pytest.mark.parametrize("split", [0, 4])
@pytest.mark.parametrize("k,cin,cout,h,w", [(5, 224, 128, 16, 16), (5, 48, 32, 16, 24), (1, 96, 64, 32, 40), (3, 64, 96, 8, 12)])
def test_conv_checkerboard_output(k, cin, cout, h, w, split):
    """ConvArgs::ckbd (entropy-parameter nets, engine.hip entropy_params): only one checkerboard half of the output is
    computed; those values are bit-identical to the full convolution, the other half is left alone (reads 0 here)."""
    dev = require_gpu()
    from rgbd_amd._lib import check, lib

    g = torch.Generator().manual_seed(k * 100 + cin)
    x = torch.randn(3, cin, h, w, generator=g)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)).contiguous()
    b = torch.randn(cout, generator=g)
    f32p = ctypes.POINTER(ctypes.c_float)

    def run():
        xd = x.to(dev).contiguous()
        yd = torch.empty((3, cout, h, w), device=dev)
        check(lib().rgbd_conv2d_nchw(ctypes.c_void_p(xd.data_ptr()), 3, cin, h, w, wt.numpy().ctypes.data_as(f32p),
                                     b.numpy().ctypes.data_as(f32p), cout, k, 1, k // 2, 0, 1, None,
                                     ctypes.c_void_p(yd.data_ptr()), None), "conv2d")
        return yd.cpu()

    lib().rgbd_debug_force_splitk(split)
    try:
        full = run()
        yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        for part in (1, 2):
            lib().rgbd_debug_force_ckbd(part)
            got = run()
            keep = ((yy + xx) % 2 == (1 if part == 1 else 0))  # anchor: (row + col) odd (ckbd.py:37-48)
            assert torch.equal(got[..., keep], full[..., keep])
            assert got[..., ~keep].abs().max().item() == 0.0
    finally:
        lib().rgbd_debug_force_ckbd(0)
        lib().rgbd_debug_force_splitk(0)


@pytest.mark.gpu
@pytest.mark.parametrize("tile", ["2,3,8,16,1", "2,2,8,16,0", "1,3,4,16,1", "2,5,4,16,0", "1,1,1,16,1", "2,3,4,64,0",
                                  "2,2,4,16,4", "2,3,4,16,4", "2,5,2,16,4", "1,3,2,16,4", "1,1,1,16,4", "2,2,8,16,5", "1,3,4,16,5"])
def test_tile_choice_never_changes_a_bit(tile):
    """The determinism contract behind the tile table / cost model: every tile shape, stage depth and
    staging mode produces the same fp32 bits, so tile selection is a pure speed matter (256-pixel tiles included).
    Staging modes 4 / 5 are the ring of four / three DMA stage buffers of the single-tap layers (round 4); the baseline is
    a register-staged tile, whatever the cost model would pick."""
    dev = require_gpu()
    from rgbd_amd._lib import check, lib

    f32p = ctypes.POINTER(ctypes.c_float)
    shapes = [(2, 96, 32, 48, 96, 3, 1, 1, 0), (1, 192, 40, 24, 96, 1, 1, 0, 0), (1, 64, 16, 24, 96, 5, 2, 2, 1),
              (1, 64, 33, 47, 80, 3, 1, 1, 0), (2, 1344, 9, 13, 213, 1, 1, 0, 0), (3, 32, 21, 17, 469, 1, 1, 0, 0),
              (1, 16, 8, 8, 16, 1, 1, 0, 0)]
    launched = 0
    for n, cin, h, w, cout, k, s, p, tr in shapes:
        g = torch.Generator().manual_seed(n + cin + h + w + cout)
        x = torch.randn(n, cin, h, w, generator=g).to(dev)
        wt = (torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) / (k * cin ** 0.5)).contiguous()
        b = torch.randn(cout, generator=g)
        oh = (h - 1) * s - 2 * p + k + (s - 1) if tr else (h + 2 * p - k) // s + 1
        ow = (w - 1) * s - 2 * p + k + (s - 1) if tr else (w + 2 * p - k) // s + 1

        def run():
            y = torch.empty((n, cout, oh, ow), device=dev)
            rc = lib().rgbd_conv2d_nchw(ctypes.c_void_p(x.data_ptr()), n, cin, h, w, wt.numpy().ctypes.data_as(f32p),
                                        b.numpy().ctypes.data_as(f32p), cout, k, s, p, tr, 1, None,
                                        ctypes.c_void_p(y.data_ptr()), None)
            return rc, y.cpu()

        lib().rgbd_debug_force_tile(b"2,2,2,16,0")
        try:
            rc0, y0 = run()
        finally:
            lib().rgbd_debug_force_tile(b"")
        check(rc0, "conv2d")
        lib().rgbd_debug_force_tile(tile.encode())
        try:
            rc1, y1 = run()
        finally:
            lib().rgbd_debug_force_tile(b"")
        if rc1 == -28:  # this tile cannot hold the layer's patch / taps: the launcher refuses it, nothing to compare
            continue
        check(rc1, "conv2d (forced tile)")
        assert torch.equal(y0, y1), (tile, (n, cin, h, w, cout, k, s, p, tr))
        launched += 1
    assert launched >= 1  # (a 64-channel stage only exists for layers whose taps all fit one stage)


@pytest.mark.gpu
@pytest.mark.parametrize("cout,h,w", [(3, 16, 24), (1, 16, 16), (3, 9, 13), (4, 32, 40)])
def test_subpixel_deconv(cout, h, w):
    """The image-producing ConvTranspose2d (N -> 3 / 1, k 5, stride 2) as one 9-tap sub-pixel conv over the input grid:
    the same bits as the four-phase form (zero weights only add fma(0, x, acc) == acc steps to each chain), and the
    torch result within the kernel's usual tolerance."""
    dev = require_gpu()
    from rgbd_amd._lib import check, lib

    g = torch.Generator().manual_seed(100 + cout + h)
    x = torch.randn(2, 192, h, w, generator=g)
    wt = (torch.randn(192, cout, 5, 5, generator=g) / 40).contiguous()
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv_transpose2d(x, wt, b, stride=2, padding=2, output_padding=1)
    f32p = ctypes.POINTER(ctypes.c_float)
    xd = x.to(dev).contiguous()

    def run(mode):
        check(lib().rgbd_debug_force_subpix(mode), "force_subpix")
        yd = torch.empty(ref.shape, device=dev)
        check(lib().rgbd_conv2d_nchw(ctypes.c_void_p(xd.data_ptr()), 2, 192, h, w, wt.numpy().ctypes.data_as(f32p),
                                     b.numpy().ctypes.data_as(f32p), cout, 5, 2, 2, 1, 0, None, ctypes.c_void_p(yd.data_ptr()),
                                     None), "conv2d")
        return yd.cpu()

    try:
        phased, packed = run(0), run(2)
    finally:
        lib().rgbd_debug_force_subpix(1)
    assert torch.equal(phased, packed)
    assert (packed - ref).abs().max().item() <= 2e-5 * (ref.abs().max().item() + 1e-3)
