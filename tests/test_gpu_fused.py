"""Fused ResidualBottleneck / ResidualUnit tails (3x3 + ReLU -> 1x1 + residual in one launch, csrc/conv_mfma.hip): the
intermediate stays in the accumulator registers and feeds the second GEMM in the channel order the stand-alone 1x1
kernel uses, and the next block's leading 1x1 can ride along as a third GEMM fed from the finished output groups; every
mode -- never fused, fused with 64 / 128 / 256-pixel tiles, with or without the leading layer -- must give the same bits:
latents, streams and reconstructions are compared with array_equal, on a ragged map (24 columns under 16-wide tiles) and
on a batch."""
import numpy as np
import pytest
import torch

from gpu_utils import require_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net(synth_sd):
    require_gpu()
    import rgbd_amd

    m = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
    m.load_state_dict(synth_sd)
    m.update(force=True)
    return m.to("cuda")


def _run(net, r, d):
    out = net.compress(r, d)
    lat = {k: net.debug_tensor(k).copy() for k in ("y_r", "y_d", "yhat_r", "yhat_d")}
    rec = net.decompress(out["r_strings"], out["d_strings"], out["shape"])
    return out, lat, rec["x_hat"]["r"].cpu().numpy(), rec["x_hat"]["d"].cpu().numpy()


@pytest.mark.parametrize("shape", [(1, 128, 192), (2, 128, 128), (1, 256, 384)], ids=str)
def test_fused_tail_bit_identical(net, shape):
    from rgbd_amd import synth
    from rgbd_amd._lib import check, lib

    B, H, W = shape
    r, d = synth.synthetic_batch(B, H, W, config_id=7)
    r, d = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
    net.per_image_streams = True
    try:
        check(lib().rgbd_debug_force_fuse(0), "force_fuse")
        ref = _run(net, r, d)
        for mode in (1, 2, 4, -1, 15, 17, 18, 20):
            check(lib().rgbd_debug_force_fuse(mode), "force_fuse")
            got = _run(net, r, d)
            for k in ref[1]:
                assert np.array_equal(ref[1][k], got[1][k]), (mode, k)
            assert ref[0]["r_strings"] == got[0]["r_strings"] and ref[0]["d_strings"] == got[0]["d_strings"], mode
            assert np.array_equal(ref[2], got[2]) and np.array_equal(ref[3], got[3]), mode
    finally:
        lib().rgbd_debug_force_fuse(-1)
        net.per_image_streams = False


def test_fused_tail_is_taken(net):
    """The automatic plan fuses the large maps of a 2 x 256 x 320 batch: fewer stand-alone conv launches with the tails
    fused, fewer again with the following blocks' leading layers riding along."""
    from rgbd_amd import synth
    from rgbd_amd._lib import check, lib

    r, d = synth.synthetic_batch(2, 256, 320, config_id=3)
    r, d = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
    counts = {}
    try:
        for mode in (0, 15, -1):
            check(lib().rgbd_debug_force_fuse(mode), "force_fuse")
            check(lib().rgbd_debug_conv_log(1), "conv_log")
            net.compress(r, d)
            torch.cuda.synchronize()
            need = lib().rgbd_debug_conv_log_read(None, 0)
            import ctypes

            buf = ctypes.create_string_buffer(int(need))
            lib().rgbd_debug_conv_log_read(buf, need)
            lib().rgbd_debug_conv_log(0)
            counts[mode] = sum(int(line.rsplit(",", 1)[1]) for line in buf.value.decode().splitlines()[1:] if line)
    finally:
        lib().rgbd_debug_force_fuse(-1)
        lib().rgbd_debug_conv_log(0)
    assert counts[-1] < counts[15] < counts[0], counts


def test_image_layers_packed_forms_same_bits(net):
    """The codec's first conv over its K-packed input and its last transposed conv in sub-pixel form (defaults) against
    the tap-by-tap / four-phase forms (rgbd_debug_force_subpix(0) selects both): same latents, streams, reconstructions."""
    from rgbd_amd import synth
    from rgbd_amd._lib import check, lib

    r, d = synth.synthetic_batch(2, 128, 192, config_id=11)
    r, d = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
    try:
        check(lib().rgbd_debug_force_subpix(0), "force_subpix")
        ref = _run(net, r, d)
        check(lib().rgbd_debug_force_subpix(1), "force_subpix")
        got = _run(net, r, d)
    finally:
        lib().rgbd_debug_force_subpix(1)
    for k in ref[1]:
        assert np.array_equal(ref[1][k], got[1][k]), k
    assert np.array_equal(ref[2], got[2]) and np.array_equal(ref[3], got[3])
    assert ref[0]["r_strings"] == got[0]["r_strings"] and ref[0]["d_strings"] == got[0]["d_strings"]
