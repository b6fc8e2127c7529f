"""HIP-graph replay of compress() / decompress(): the first call of a shape runs eagerly, the second is captured, later
ones are replayed -- all four must give the same bytes and the same pixels, across workspace growth (a larger shape in
between re-allocates the arena and drops the cached graphs), tile-mode switches and new inputs."""
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


def _pair(B, H, W, cid):
    from rgbd_amd import synth

    r, d = synth.synthetic_batch(B, H, W, config_id=cid)
    return torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()


def _round(net, r, d):
    out = net.compress(r, d)
    yhat = net.debug_tensor("yhat_r").copy()
    rec = net.decompress(out["r_strings"], out["d_strings"], out["shape"])
    assert np.array_equal(net.debug_tensor("yhat_r"), yhat)
    return out, rec["x_hat"]["r"].clone(), rec["x_hat"]["d"].clone()


@pytest.mark.parametrize("side_stream", [False, True])
def test_eager_capture_replay_agree(net, side_stream):
    """On a torch side stream the second call of a shape is captured and later ones are replayed; on the caller's NULL
    stream (which cannot be captured) every call takes the eager path -- same bytes, same pixels either way."""
    if side_stream:
        with torch.cuda.stream(torch.cuda.Stream()):
            _eager_capture_replay(net)
        torch.cuda.synchronize()
    else:
        _eager_capture_replay(net)


def _eager_capture_replay(net):
    net.per_image_streams = True
    try:
        a = _pair(2, 128, 192, 31)
        b = _pair(2, 128, 192, 32)  # same shape, other pixels: a replayed graph must read the NEW inputs
        ref_a = _round(net, *a)     # eager (and sizes the workspace: a re-allocation drops cached graphs)
        cap_a = _round(net, *a)     # eager or captured + launched, depending on when the workspace last grew
        _round(net, *a)
        if torch.cuda.current_stream().cuda_stream:
            assert net.graph_count() >= 2  # compress and decompress of this shape are cached graphs now
        rep_a = _round(net, *a)     # replayed
        rep_b = _round(net, *b)     # replayed on other inputs
        for other in (cap_a, rep_a):
            assert other[0]["r_strings"] == ref_a[0]["r_strings"] and other[0]["d_strings"] == ref_a[0]["d_strings"]
            assert torch.equal(other[1], ref_a[1]) and torch.equal(other[2], ref_a[2])
        assert rep_b[0]["r_strings"] != ref_a[0]["r_strings"]
        # a larger shape grows the workspace: every cached graph is dropped, results stay the same afterwards
        big = _pair(1, 256, 320, 33)
        _round(net, *big)
        again_b = _round(net, *b)   # eager again (fresh cache)
        cap_b = _round(net, *b)
        rep_b2 = _round(net, *b)
        for other in (again_b, cap_b, rep_b2):
            assert other[0]["r_strings"] == rep_b[0]["r_strings"] and other[0]["d_strings"] == rep_b[0]["d_strings"]
            assert torch.equal(other[1], rep_b[1]) and torch.equal(other[2], rep_b[2])
        # tile mode is part of the graph key; outputs are bit-identical in both modes
        net.set_tile_mode("throughput")
        for _ in range(3):
            t = _round(net, *b)
            assert t[0]["r_strings"] == rep_b[0]["r_strings"] and torch.equal(t[1], rep_b[1])
        net.set_tile_mode("latency")
        # a decoder fed a stream that does not fit the shape's slot is refused, not overrun
        with pytest.raises(Exception):
            bad = [[rep_b[0]["r_strings"][0][0] * 40] * 2, rep_b[0]["r_strings"][1]]
            net.decompress(bad, rep_b[0]["d_strings"], rep_b[0]["shape"])
    finally:
        net.per_image_streams = False
        net.set_tile_mode("latency")


def test_bicee_alone_replay(net):
    from rgbd_amd import synth

    yr, hr, yd, hd = [torch.from_numpy(a).cuda() for a in synth.synthetic_latents(1, 16, 16, 320, 5)]
    outs = []
    for _ in range(3):
        sr, sd_ = net.compress_united(yr, hr, yd, hd)
        yh_r, yh_d = net.decompress_united(sr[0], hr, sd_[0], hd)
        outs.append((sr, sd_, yh_r.clone(), yh_d.clone()))
    for o in outs[1:]:
        assert o[0] == outs[0][0] and o[1] == outs[0][1]
        assert torch.equal(o[2], outs[0][2]) and torch.equal(o[3], outs[0][3])


def test_stf_replay_and_varying_shapes():
    """STF_united zero-fills y_hat inside the captured body (its 24-wide slices make a 16-channel read straddle into a slice
    that is not coded yet): a replayed graph must do that fill too.  Shapes alternate so that the workspace holds another
    call's leftovers whenever a graph is replayed."""
    import rgbd_amd
    from rgbd_amd import synth

    require_gpu()
    m = rgbd_amd.modelZoo["STF_united"](config=rgbd_amd.model_config(), channel=4).eval()
    m.load_state_dict(synth.synthetic_state_dict(0, model="STF_united"))
    m.update(force=True)
    m = m.to("cuda")
    shapes = [(1, 256, 256, 41), (1, 256, 320, 42)]
    ref = {}
    with torch.cuda.stream(torch.cuda.Stream()):  # (the NULL stream cannot be captured)
        for rnd in range(4):  # eager, capture, replay, replay -- interleaved over two shapes
            for shp in shapes:
                r, d = _pair(*shp)
                out = m.compress(r, d)
                rec = m.decompress(out["r_strings"], out["d_strings"], out["shape"])
                got = (out["r_strings"], out["d_strings"], rec["x_hat"]["r"].clone(), rec["x_hat"]["d"].clone())
                if shp not in ref:
                    ref[shp] = got
                else:
                    assert got[0] == ref[shp][0] and got[1] == ref[shp][1], (rnd, shp)
                    assert torch.equal(got[2], ref[shp][2]) and torch.equal(got[3], ref[shp][3]), (rnd, shp)
        assert m.graph_count() >= 4
    torch.cuda.synchronize()
