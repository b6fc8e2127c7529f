"""RGB / depth layer pairs as ONE grouped launch (ConvArgs::groups == 2; rgbd_debug_force_pair): the two branches of g_a, g_s,
h_a, h_s and the per-slice channel-context nets run the same layer shapes on independent data.  A grouped launch keeps
every output's fma chain, so streams, latents and reconstructions must equal the two-launch form bit for bit -- pairing is
a speed decision like the tile choice (reference: modules/transform/analysis.py:116-174, synthesis.py:126-184,305-323,
models/elic_united.py:288-333)."""
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


def _roundtrip(net, rgb, depth):
    out = net.compress(rgb, depth)
    yh = [net.debug_tensor("yhat_r").copy(), net.debug_tensor("yhat_d").copy()]
    rec = net.decompress(out["r_strings"], out["d_strings"], out["shape"])
    return out, yh, rec["x_hat"]["r"].clone(), rec["x_hat"]["d"].clone()


@pytest.mark.parametrize("B,H,W,cid,tiles", [(1, 128, 192, 31, "latency"), (2, 256, 256, 32, "throughput"),
                                             (1, 512, 640, 33, "latency")])
def test_grouped_launches_same_bits(net, B, H, W, cid, tiles):
    from rgbd_amd import synth
    from rgbd_amd._lib import check, lib

    r, d = synth.synthetic_batch(B, H, W, config_id=cid)
    rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
    net.per_image_streams = True
    net.set_tile_mode(tiles)
    try:
        check(lib().rgbd_debug_force_pair(0), "force_pair")
        ref = _roundtrip(net, rgb, depth)
        net.set_profile(True)
        _roundtrip(net, rgb, depth)
        n_single = net.profile_read()["launches"]
        net.set_profile(False)
        check(lib().rgbd_debug_force_pair(1), "force_pair")
        got = _roundtrip(net, rgb, depth)
        net.set_profile(True)
        _roundtrip(net, rgb, depth)
        n_pair = net.profile_read()["launches"]
        net.set_profile(False)
    finally:
        check(lib().rgbd_debug_force_pair(1), "force_pair")
        net.per_image_streams = False
        net.set_tile_mode("latency")
    assert got[0]["r_strings"] == ref[0]["r_strings"] and got[0]["d_strings"] == ref[0]["d_strings"]
    assert np.array_equal(got[1][0], ref[1][0]) and np.array_equal(got[1][1], ref[1][1])
    assert torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3])
    # the transforms' and the channel-context nets' launches halve (the four entropy-parameter nets per slice, the local
    # context convs and the image-facing first conv stay single: they depend on each other / differ in shape)
    print(f"conv launches per enc+dec: {n_single} as single launches, {n_pair} with grouped pairs")
    assert n_pair < 0.75 * n_single


def test_grouped_launches_eval_forward(net):
    from rgbd_amd import synth
    from rgbd_amd._lib import check, lib

    r, d = synth.synthetic_batch(1, 128, 128, config_id=34)
    rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
    try:
        check(lib().rgbd_debug_force_pair(0), "force_pair")
        a = net(rgb, depth)
        check(lib().rgbd_debug_force_pair(1), "force_pair")
        b = net(rgb, depth)
    finally:
        check(lib().rgbd_debug_force_pair(1), "force_pair")
    assert torch.equal(a["x_hat"]["r"], b["x_hat"]["r"]) and torch.equal(a["x_hat"]["d"], b["x_hat"]["d"])
    assert torch.equal(a["r_likelihoods"]["y"], b["r_likelihoods"]["y"]) and torch.equal(a["d_likelihoods"]["z"], b["d_likelihoods"]["z"])
