"""GPU rANS coder (through the C ABI / the compressai.ans look-alike classes) vs the oracle coder: bit-exact."""
import numpy as np
import pytest

from gpu_utils import require_gpu
from oracle import coder

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_tables(kat):
    require_gpu()
    from rgbd_amd import ans

    return ans.Tables(kat["gc_cdf"], kat["gc_sizes"], kat["gc_offsets"])


def test_tiny_and_b2(kat, gc_tables, gpu_tables):
    from rgbd_amd import ans

    s = ans._encode(gpu_tables, kat["tiny_sym"], kat["tiny_idx"])
    assert s == kat["tiny_stream"].tobytes()
    rng = np.random.RandomState(1234)
    n = 49152
    idx = rng.randint(0, 64, n)
    sym = np.rint(rng.standard_normal(n) * kat["scale_table"][idx]).astype(np.int64)
    sym[::97] *= 8
    sym[5::193] = -sym[5::193] - 3
    sym, idx = sym.astype(np.int32), idx.astype(np.int32)
    s = ans._encode(gpu_tables, sym, idx)
    assert s == kat["b2_stream"].tobytes()
    assert s == coder.rans_encode(sym, idx, gc_tables)


def test_list_api_like_reference(kat):
    require_gpu()
    from rgbd_amd import ans

    cdf_l, sz_l, off_l = kat["gc_cdf"].tolist(), kat["gc_sizes"].tolist(), kat["gc_offsets"].tolist()
    sym, idx = kat["tiny_sym"].tolist(), kat["tiny_idx"].tolist()
    want = kat["tiny_stream"].tobytes()
    assert ans.RansEncoder().encode_with_indexes(sym, idx, cdf_l, sz_l, off_l) == want
    enc = ans.BufferedRansEncoder()
    enc.encode_with_indexes(sym[:5], idx[:5], cdf_l, sz_l, off_l)
    enc.encode_with_indexes(sym[5:], idx[5:], cdf_l, sz_l, off_l)
    assert enc.flush() == want
    dec = ans.RansDecoder()
    dec.set_stream(want)
    got = dec.decode_stream(idx[:4], cdf_l, sz_l, off_l) + dec.decode_stream(idx[4:9], cdf_l, sz_l, off_l) \
        + dec.decode_stream(idx[9:], cdf_l, sz_l, off_l)
    assert got == sym
    assert ans.RansDecoder().decode_with_indexes(want, idx, cdf_l, sz_l, off_l) == sym
    assert ans.pmf_to_quantized_cdf([0.1, 0.2, 0.7], 16) == [0, 6554, 19661, 65536]


@pytest.mark.parametrize("seed,n", [(0, 1), (1, 2), (2, 511), (3, 512), (4, 513), (5, 30000), (6, 200000)])
def test_random_roundtrip_vs_oracle(seed, n, kat, gc_tables, gpu_tables):
    from rgbd_amd import ans

    rng = np.random.RandomState(seed)
    idx = rng.randint(0, 64, n).astype(np.int32)
    sym = np.rint(rng.standard_normal(n) * kat["scale_table"][idx] * (1 + 3 * (rng.rand(n) < 0.02))).astype(np.int32)
    esc = rng.rand(n) < 0.01
    sym[esc] = rng.randint(-100000, 100000, int(esc.sum()))
    s = ans._encode(gpu_tables, sym, idx)
    assert s == coder.rans_encode(sym, idx, gc_tables)
    import ctypes

    from rgbd_amd._lib import check, lib

    d = ans.RansDecoder()
    d.set_stream(s)
    cuts = sorted(set([0, n // 3, n // 2, n]))
    got = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        out = np.empty(b - a, dtype=np.int32)
        ii = np.ascontiguousarray(idx[a:b])
        check(lib().rgbd_rans_decoder_decode(d._h, gpu_tables.handle, ii.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                             b - a, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))), "decode")
        got.append(out)
    assert np.array_equal(np.concatenate(got), sym)


def test_empty_and_escape_only(kat, gc_tables, gpu_tables):
    from rgbd_amd import ans

    assert ans._encode(gpu_tables, [], []) == (1 << 31).to_bytes(8, "little")
    s = ans._encode(gpu_tables, kat["esc_sym"], np.zeros_like(kat["esc_sym"]))
    assert s == kat["esc_stream"].tobytes()


def test_bottleneck_tables(kat):
    require_gpu()
    from rgbd_amd import ans

    t = ans.Tables(kat["rgb_eb_cdf"], kat["rgb_eb_sizes"], kat["rgb_eb_offsets"])
    ot = coder.Tables(kat["rgb_eb_cdf"], kat["rgb_eb_sizes"], kat["rgb_eb_offsets"])
    rng = np.random.RandomState(7)
    idx = np.repeat(np.arange(192, dtype=np.int32), 24)
    sym = np.rint(rng.standard_normal(idx.shape[0]) * 4).astype(np.int32)
    sym[::50] = 60
    s = ans._encode(t, sym, idx)
    assert s == coder.rans_encode(sym, idx, ot)
    assert np.array_equal(coder.rans_decode(s, idx, ot), sym)


@pytest.mark.parametrize("esc_rate,n", [(0.2, 60000), (0.6, 20000), (1.0, 5000)])
def test_escape_heavy_streams(esc_rate, n, kat, gc_tables, gpu_tables):
    """Escapes of every payload length the reference can code (1..7 nibbles: its nibble-count loop is undefined for raw
    values >= 2^28, rans_interface.cpp:143-145), both signs, at rates that keep the ISA escape paths, their window /
    block-boundary fall-backs and the generic paths all busy."""
    from rgbd_amd import ans

    rng = np.random.RandomState(int(esc_rate * 10) + n)
    idx = rng.randint(0, 64, n).astype(np.int32)
    sym = np.rint(rng.standard_normal(n) * kat["scale_table"][idx]).astype(np.int64)
    esc = rng.rand(n) < esc_rate
    mag = (2.0 ** rng.uniform(0, 26.5, int(esc.sum()))).astype(np.int64)
    sym[esc] = np.where(rng.rand(int(esc.sum())) < 0.5, mag, -mag)
    sym = sym.astype(np.int32)
    s = ans._encode(gpu_tables, sym, idx)
    assert s == coder.rans_encode(sym, idx, gc_tables)
    d = ans.RansDecoder()
    d.set_stream(s)
    assert np.array_equal(np.asarray(d.decode_stream(idx, kat["gc_cdf"], kat["gc_sizes"], kat["gc_offsets"]), np.int32), sym)


def test_very_wide_row():
    """A row with 20,000 symbols: more than 64 candidates per bucket, so the decoder's 64-lane probe has to continue
    (generic path), next to a narrow row in the same table."""
    require_gpu()
    from rgbd_amd import ans

    n_sym = 20000
    freq = np.full(n_sym + 1, 3, np.int64)  # last entry = escape slot
    freq[: 65536 - int(freq.sum())] += 1
    wide = np.concatenate([[0], np.cumsum(freq)]).astype(np.int32)
    assert wide[-1] == 65536
    cdf = np.zeros((2, wide.shape[0]), np.int32)
    cdf[0] = wide
    cdf[1, :5] = [0, 20000, 40000, 65000, 65536]
    sizes = np.array([wide.shape[0], 5], np.int32)
    offsets = np.array([-10000, -1], np.int32)
    t, ot = ans.Tables(cdf, sizes, offsets), coder.Tables(cdf, sizes, offsets)
    rng = np.random.RandomState(3)
    n = 40000
    idx = (rng.rand(n) < 0.2).astype(np.int32)
    sym = np.where(idx == 0, rng.randint(-10050, 10050, n), rng.randint(-3, 5, n)).astype(np.int32)
    s = ans._encode(t, sym, idx)
    assert s == coder.rans_encode(sym, idx, ot)
    d = ans.RansDecoder()
    d.set_stream(s)
    assert np.array_equal(np.asarray(d.decode_stream(idx, cdf, sizes, offsets), np.int32), sym)


@pytest.mark.parametrize("seed", [0, 1])
def test_rows_around_the_lane_count(seed):
    """Rows of 2 ... 70 slots with random frequencies (many of them 1): the decoder's first level holds 64 slots of a row in
    the 64 lanes, so 63 / 64 / 65 slots are its edges (escape slot in the last lane; slot 63 standing for the rest of a wider
    row), and symbol 0 / the slot before the escape are the edges of the lane search.  Every slot of every row is coded, with
    escapes of both signs in between and stream lengths around the 64-symbol batch."""
    require_gpu()
    from rgbd_amd import ans

    from coder_cases import lane_edge_symbols, lane_edge_tables

    cdf, sizes, offsets, rng = lane_edge_tables(seed)
    t, ot = ans.Tables(cdf, sizes, offsets), coder.Tables(cdf, sizes, offsets)
    for n in (1, 63, 64, 65, 4000):
        idx, sym = lane_edge_symbols(rng, n, sizes, offsets)
        s = ans._encode(t, sym, idx)
        assert s == coder.rans_encode(sym, idx, ot)
        d = ans.RansDecoder()
        d.set_stream(s)
        assert np.array_equal(np.asarray(d.decode_stream(idx, cdf, sizes, offsets), np.int32), sym), n


@pytest.mark.parametrize("seed", [0, 1])
def test_rows_around_the_coarse_level(seed):
    """Rows of 129 ... 4032 slots have a coarse first level in the decoder (blocks of ceil(slots / 64) symbols, then one
    64-wide probe of the block); a batch of 64 symbols takes that path when at least five of its symbols sit on such rows.
    Every slot of every row is coded -- block edges, symbol 0, the slot before the escape, escapes of both signs -- in
    mixes from "hardly any symbol on such a row" (the bucket-table path) to "all of them", next to narrow rows, to wide
    rows without a coarse level (65 ... 128 and 4033 slots) and around the batch length."""
    require_gpu()
    from rgbd_amd import ans

    from coder_cases import COARSE_EDGE_SLOTS, lane_edge_symbols, lane_edge_tables

    cdf, sizes, offsets, rng = lane_edge_tables(seed, COARSE_EDGE_SLOTS)
    t, ot = ans.Tables(cdf, sizes, offsets), coder.Tables(cdf, sizes, offsets)
    coarse = [r for r, n in enumerate(COARSE_EDGE_SLOTS) if 128 < n <= 4032]
    other = [r for r in range(len(COARSE_EDGE_SLOTS)) if r not in coarse]
    for n, share in ((63, 1.0), (64, 1.0), (65, 0.5), (4000, 0.03), (4000, 0.1), (4000, 0.5), (20000, 1.0)):
        idx, sym = lane_edge_symbols(rng, n, sizes, offsets, rows=coarse)
        idx2, sym2 = lane_edge_symbols(rng, n, sizes, offsets, rows=other)
        pick = rng.rand(n) < share
        idx, sym = np.where(pick, idx, idx2).astype(np.int32), np.where(pick, sym, sym2).astype(np.int32)
        s = ans._encode(t, sym, idx)
        assert s == coder.rans_encode(sym, idx, ot)
        d = ans.RansDecoder()
        d.set_stream(s)
        assert np.array_equal(np.asarray(d.decode_stream(idx, cdf, sizes, offsets), np.int32), sym), (n, share)


def test_both_encoder_generations_make_the_same_stream(kat, gc_tables, gpu_tables):
    """The library keeps its first encoder loop selectable (RGBD_CODER_V1, read once per process) for A/B timing.  Both
    generations -- the older one in its own process -- must produce the oracle's bytes for a stream with escapes of every
    length and a ragged last batch."""
    import hashlib
    import os
    import subprocess
    import sys

    from rgbd_amd import ans

    rng = np.random.RandomState(11)
    n = 20011
    idx = rng.randint(0, 64, n).astype(np.int32)
    sym = np.rint(rng.standard_normal(n) * kat["scale_table"][idx]).astype(np.int64)
    esc = rng.rand(n) < 0.15
    mag = (2.0 ** rng.uniform(0, 26.5, int(esc.sum()))).astype(np.int64)
    sym[esc] = np.where(rng.rand(int(esc.sum())) < 0.5, mag, -mag)
    sym = sym.astype(np.int32)
    want = hashlib.sha256(coder.rans_encode(sym, idx, gc_tables)).hexdigest()
    assert hashlib.sha256(ans._encode(gpu_tables, sym, idx)).hexdigest() == want
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r); import rgbd_amd; from rgbd_amd import ans;"
            "from rgbd_amd.entropy_models import GaussianConditional, get_scale_table;"
            "gc = GaussianConditional(); gc.update_scale_table(get_scale_table(), force=True);"
            "t = ans.Tables(*gc.numpy_tables()); d = np.load(sys.argv[1]);"
            "print(hashlib.sha256(ans._encode(t, d['sym'], d['idx'])).hexdigest())" % root)
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "in.npz")
        np.savez(path, sym=sym, idx=idx)
        for var in ("RGBD_CODER_V1",):
            env = dict(os.environ, **{var: "1"})
            out = subprocess.run([sys.executable, "-c", prog, path], env=env, capture_output=True, text=True, timeout=120)
            assert out.returncode == 0, out.stderr[-2000:]
            assert out.stdout.split()[-1] == want, var


# ---- the operator-level exports of SURVEY 8(b) row 3: device-resident, many streams per launch ------------------------------
def _vp(t):
    import ctypes

    return ctypes.c_void_p(t.data_ptr())


def test_device_batched_coder_streams_equal_the_oracle_and_decode_in_parts(kat, gc_tables, gpu_tables):
    """rgbd_rans_encode_batch_dev / rgbd_rans_decode_batch_dev: 7 ragged streams (one EMPTY) in one launch each; every stream byte
    for byte what the oracle's single-stream coder (rans_interface.cpp:99-205) produces; decoded in three calls that continue
    from the kept state (decode_stream after decode_stream, :286-351)."""
    dev = require_gpu()
    import torch

    from rgbd_amd._lib import check, lib

    rng = np.random.RandomState(77)
    counts = [3000, 1, 0, 4097, 12000, 640, 2999]
    step = 600  # every non-empty stream is decoded `step` symbols per call where it has them: use equal-length prefixes below
    base = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    n = int(base[-1])
    idx = rng.randint(0, 64, n).astype(np.int32)
    sym = np.rint(rng.standard_normal(n) * kat["scale_table"][idx] * (1 + 3 * (rng.rand(n) < 0.02))).astype(np.int32)
    esc = rng.rand(n) < 0.01
    sym[esc] = rng.randint(-50000, 50000, int(esc.sum()))
    d_sym = torch.from_numpy(np.concatenate([sym, [0]]).astype(np.int32)).to(dev)  # (readable one element past the end)
    d_idx = torch.from_numpy(np.concatenate([idx, [0]]).astype(np.int32)).to(dev)
    d_base = torch.from_numpy(base[:-1].copy()).to(dev)
    d_cnt = torch.tensor(counts, dtype=torch.int64, device=dev)
    ns = len(counts)
    cap = int(lib().rgbd_rans_max_bytes(max(counts))) // 4
    assert cap % 64 == 0
    d_out = torch.zeros(ns * cap, dtype=torch.int32, device=dev)
    d_words = torch.zeros(ns, dtype=torch.int64, device=dev)
    d_err = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    check(lib().rgbd_rans_encode_batch_dev(gpu_tables.handle, _vp(d_sym), _vp(d_idx), _vp(d_base), _vp(d_cnt), ns, _vp(d_out), cap,
                                           _vp(d_words), _vp(d_err), stream.cuda_stream), "encode_batch_dev")
    stream.synchronize()
    assert int(d_err.item()) == 0
    words = d_words.cpu().numpy()
    out = d_out.cpu().numpy().view(np.uint32).reshape(ns, cap)
    streams = []
    for s_ in range(ns):
        got = out[s_, cap - words[s_]:].tobytes()
        want = coder.rans_encode(sym[base[s_]:base[s_ + 1]], idx[base[s_]:base[s_ + 1]], gc_tables)
        assert got == want, f"stream {s_} ({counts[s_]} symbols)"
        streams.append(got)
    # decode: the streams back to back in one buffer; `count` symbols per stream and call, so streams of one length share a call
    blob = np.frombuffer(b"".join(streams), dtype=np.uint32)
    lens = np.array([len(s_) // 4 for s_ in streams], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    d_blob = torch.from_numpy(blob.view(np.int32).copy()).to(dev)
    got_sym = torch.full((n + 1,), -12345, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):  # the small argument tensors are allocated, filled and freed in the order of THIS stream
        for s_ in range(ns):
            if not counts[s_]:
                continue
            d_off, d_len = torch.tensor([offs[s_]], device=dev), torch.tensor([lens[s_]], device=dev)
            d_sb = torch.tensor([base[s_]], dtype=torch.int64, device=dev)
            state = torch.zeros(2, dtype=torch.int64, device=dev)
            cuts = sorted(set([0, min(step, counts[s_]), counts[s_] // 2, counts[s_]]))
            for a, b in zip(cuts[:-1], cuts[1:]):
                check(lib().rgbd_rans_decode_batch_dev(gpu_tables.handle, _vp(d_blob), _vp(d_off), _vp(d_len), 1, _vp(state),
                                                       1 if a == 0 else 0, _vp(d_idx), _vp(got_sym), _vp(d_sb), a, b - a,
                                                       stream.cuda_stream), "decode_batch_dev")
    stream.synchronize()
    assert np.array_equal(got_sym.cpu().numpy()[:n], sym)
    # ... and several equal-length streams in ONE decode launch (what the codec does for the images of a batch)
    k = 2048
    eq = rng.randint(0, 64, (5, k)).astype(np.int32)
    es = np.rint(rng.standard_normal((5, k)) * kat["scale_table"][eq]).astype(np.int32)
    ss = [coder.rans_encode(es[i], eq[i], gc_tables) for i in range(5)]
    blob = np.frombuffer(b"".join(ss), dtype=np.uint32)
    lens = np.array([len(s_) // 4 for s_ in ss], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    with torch.cuda.stream(stream):
        d_blob = torch.from_numpy(blob.view(np.int32).copy()).to(dev)
        d_i = torch.from_numpy(eq.reshape(-1)).to(dev)
        d_s = torch.zeros(5 * k, dtype=torch.int32, device=dev)
        d_sb = torch.arange(5, dtype=torch.int64, device=dev) * k
        state = torch.zeros(10, dtype=torch.int64, device=dev)
        d_offs, d_lens = torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev)
        for part, (a, b) in enumerate([(0, 1000), (1000, k)]):
            check(lib().rgbd_rans_decode_batch_dev(gpu_tables.handle, _vp(d_blob), _vp(d_offs), _vp(d_lens), 5, _vp(state), 1 if part == 0 else 0, _vp(d_i),
                                                   _vp(d_s), _vp(d_sb), a, b - a, stream.cuda_stream), "decode_batch_dev")
            stream.synchronize()
    assert np.array_equal(d_s.cpu().numpy().reshape(5, k), es)


def test_device_batched_coder_refuses_bad_arguments(gpu_tables):
    dev = require_gpu()
    import torch

    from rgbd_amd._lib import lib

    t = torch.zeros(256, dtype=torch.int64, device=dev)
    L = lib()
    assert L.rgbd_rans_encode_batch_dev(None, _vp(t), _vp(t), _vp(t), _vp(t), 1, _vp(t), 64, _vp(t), _vp(t), None) == -22
    assert L.rgbd_rans_encode_batch_dev(gpu_tables.handle, _vp(t), _vp(t), _vp(t), _vp(t), 1, _vp(t), 100, _vp(t), _vp(t), None) == -22
    assert L.rgbd_rans_encode_batch_dev(gpu_tables.handle, None, _vp(t), _vp(t), _vp(t), 1, _vp(t), 64, _vp(t), _vp(t), None) == -22
    assert L.rgbd_rans_decode_batch_dev(gpu_tables.handle, None, _vp(t), _vp(t), 1, _vp(t), 1, _vp(t), _vp(t), _vp(t), 0, 5, None) == -22
    assert L.rgbd_rans_decode_batch_dev(gpu_tables.handle, _vp(t), _vp(t), _vp(t), 1, _vp(t), 1, _vp(t), _vp(t), _vp(t), 0, 0, None) == 0


@pytest.mark.parametrize("n,c,h,w", [(1, 16, 8, 10), (2, 48, 16, 20), (1, 5, 3, 2), (3, 160, 30, 40)])
def test_ckbd_quant_index_vs_oracle(n, c, h, w, kat):
    """rgbd_ckbd_quant_index / rgbd_ckbd_dequant against the oracle's restatement of utils/ckbd.py:37-125 +
    entropy_models.py:118-146,561-568 -- symbols, indexes (both in the reference's (n, c, h, w/2) order) and the scattered y_hat,
    bit for bit; includes scales under the 0.11 bound, exactly on table entries, and x - mean on .5 ties."""
    dev = require_gpu()
    import torch

    from oracle import elic_oracle as eo
    from rgbd_amd._lib import check, lib

    g = torch.Generator().manual_seed(n * 1000 + c)
    table = eo.scale_table()
    y = torch.randn(n, c, h, w, generator=g) * 6
    means = torch.randn(n, c, h, w, generator=g)
    scales = torch.exp(torch.randn(n, c, h, w, generator=g) * 2 - 0.5)
    scales.view(-1)[::7] = table[torch.randint(0, 64, ((scales.numel() + 6) // 7,), generator=g)]  # exactly on table entries
    scales.view(-1)[3::11] = 0.05                                                                   # under the bound
    scales.view(-1)[5::13] = -1.0
    y.view(-1)[::5] = (means.view(-1)[::5] + torch.randint(-4, 5, ((y.numel() + 4) // 5,), generator=g).float() + 0.5)  # ties
    tb = np.ascontiguousarray(table.numpy(), np.float32)
    import ctypes

    f32p = ctypes.POINTER(ctypes.c_float)
    yd, md, sd = y.to(dev), means.to(dev), scales.to(dev)
    yhat = torch.full((n, c, h, w), 7.0, device=dev)  # (the anchor half defines every position: the 7s must go)
    m = n * c * h * (w // 2)
    want_hat = torch.zeros(n, c, h, w)
    for anchor in (1, 0):
        sym = torch.zeros(m, dtype=torch.int32, device=dev)
        idx = torch.zeros(m, dtype=torch.int32, device=dev)
        check(lib().rgbd_ckbd_quant_index(_vp(yd), _vp(md), _vp(sd), n, c, h, w, anchor, tb.ctypes.data_as(f32p), _vp(sym), _vp(idx),
                                          _vp(yhat), None), "ckbd_quant_index")
        ws = eo.quantize_symbols(eo.pack(y, bool(anchor)), eo.pack(means, bool(anchor)))
        wi = eo.scale_indexes(eo.pack(scales, bool(anchor)), table)
        assert np.array_equal(sym.cpu().numpy(), ws.reshape(-1).numpy()), f"symbols, anchor={anchor}"
        assert np.array_equal(idx.cpu().numpy(), wi.reshape(-1).numpy()), f"indexes, anchor={anchor}"
        want_hat = want_hat + eo.unpack(ws.float() + eo.pack(means, bool(anchor)), bool(anchor))
        assert np.array_equal(yhat.cpu().numpy(), want_hat.numpy()), f"y_hat, anchor={anchor}"
        # the decoder's half of the step rebuilds the same y_hat from the symbols
        back = torch.full((n, c, h, w), 7.0, device=dev) if anchor else prev.clone()
        check(lib().rgbd_ckbd_dequant(_vp(sym), _vp(md), n, c, h, w, anchor, _vp(back), None), "ckbd_dequant")
        assert np.array_equal(back.cpu().numpy(), want_hat.numpy())
        prev = back
    assert lib().rgbd_ckbd_quant_index(_vp(yd), _vp(md), _vp(sd), n, c, h, w + 1, 1, tb.ctypes.data_as(f32p), _vp(sym), _vp(idx),
                                       _vp(yhat), None) == -22  # odd width: the reference's squeeze needs w even
