"""The plain-C oracle coder against (a) known answers produced by the reference's own C++ (tests/golden/coder_kat.npz)
and (b) the reference C++ itself where oracle/_ref has been built.  All integer work: bit-exact."""
import hashlib

import numpy as np
import pytest

from oracle import coder


def _sha(b):
    return hashlib.sha256(b).hexdigest()[:16]


def _b2_inputs(table):
    rng = np.random.RandomState(1234)
    n = 49152
    idx = rng.randint(0, 64, n)
    sym = np.rint(rng.standard_normal(n) * table[idx]).astype(np.int64)
    sym[::97] *= 8
    sym[5::193] = -sym[5::193] - 3
    return sym.astype(np.int32), idx.astype(np.int32)


def test_tiny_kat(kat, gc_tables):
    s = coder.rans_encode(kat["tiny_sym"], kat["tiny_idx"], gc_tables)
    assert s == kat["tiny_stream"].tobytes()
    assert s.hex() == "3ec315415cc000009f056801115a315121ffff01f7ff2f4cffffd204"
    assert np.array_equal(coder.rans_decode(s, kat["tiny_idx"], gc_tables), kat["tiny_sym"])


def test_tiny_kat_split_decode(kat, gc_tables):
    d = coder.RansDecoder()
    d.set_stream(kat["tiny_stream"].tobytes())
    idx = kat["tiny_idx"]
    got = np.concatenate([d.decode_stream(idx[:4], gc_tables), d.decode_stream(idx[4:9], gc_tables),
                          d.decode_stream(idx[9:], gc_tables)])
    assert np.array_equal(got, kat["tiny_sym"])


def test_kat_b2(kat, gc_tables):
    sym, idx = _b2_inputs(kat["scale_table"])
    assert _sha(sym.tobytes()) == kat["b2_sym_sha"].tobytes().decode()
    assert _sha(idx.tobytes()) == kat["b2_idx_sha"].tobytes().decode()
    s = coder.rans_encode(sym, idx, gc_tables)
    assert len(s) == 28664 and _sha(s) == "e74a4ba73e7caf82"
    assert s == kat["b2_stream"].tobytes()
    assert np.array_equal(coder.rans_decode(s, idx, gc_tables), sym)


def test_escape_and_short(kat, gc_tables):
    s = coder.rans_encode(kat["esc_sym"], np.zeros_like(kat["esc_sym"]), gc_tables)
    assert s == kat["esc_stream"].tobytes()
    assert np.array_equal(coder.rans_decode(s, np.zeros_like(kat["esc_sym"]), gc_tables), kat["esc_sym"])
    assert coder.rans_encode([3, -2], [20, 7], gc_tables) == kat["two_stream"].tobytes()


def test_empty_and_single_defined(gc_tables):
    # The reference under-allocates for <2 symbols (UB, rans_interface.cpp:171); the build defines them.
    e = coder.rans_encode([], [], gc_tables)
    assert e == (1 << 31).to_bytes(8, "little")
    one = coder.rans_encode([3], [20], gc_tables)
    assert len(one) == 8 and coder.rans_decode(one, [20], gc_tables).tolist() == [3]


def test_pmf_to_quantized_cdf(kat):
    for k in range(4):
        got = coder.pmf_to_quantized_cdf(kat[f"pmf{k}"], 16)
        assert np.array_equal(got, kat[f"pmf{k}_cdf"]), k
    assert coder.pmf_to_quantized_cdf([0.1, 0.2, 0.7]).tolist() == [0, 6554, 19661, 65536]


def test_gaussian_table_facts(kat):
    assert _sha(kat["gc_cdf"].astype("<i4").tobytes()) == "482d4620b3eb5802"
    assert kat["gc_sizes"][:8].tolist() == [5, 5, 5, 5, 7, 7, 7, 7]
    assert kat["gc_sizes"][-3:].tolist() == [2449, 2769, 3133]
    assert kat["gc_offsets"][-3:].tolist() == [-1223, -1383, -1565]
    assert int(kat["gc_sizes"].sum()) == 27256
    assert kat["gc_cdf"][0, :5].tolist() == [0, 1, 65534, 65535, 65536]


def test_table_construction_matches_reference(kat):
    from oracle import elic_oracle as eo

    t = eo.gaussian_tables()
    assert np.array_equal(t.cdf, kat["gc_cdf"])
    assert np.array_equal(t.sizes, kat["gc_sizes"]) and np.array_equal(t.offsets, kat["gc_offsets"])
    assert np.array_equal(eo.scale_table().numpy(), kat["scale_table"])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_against_reference_cpp(seed, kat, gc_tables):
    ref = coder.load_reference_coder()
    if ref is None:
        pytest.skip("oracle/_ref not built (make -C oracle ref needs /root/reference)")
    rng = np.random.RandomState(seed)
    n = [17, 4096, 20000][seed]
    idx = rng.randint(0, 64, n).astype(np.int32)
    sym = np.rint(rng.standard_normal(n) * kat["scale_table"][idx] * (1 + 3 * (rng.rand(n) < 0.02))).astype(np.int32)
    sym[rng.rand(n) < 0.01] = rng.randint(-100000, 100000)
    cdf_l, sz_l, off_l = gc_tables.cdf.tolist(), gc_tables.sizes.tolist(), gc_tables.offsets.tolist()
    want = ref["ans"].RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdf_l, sz_l, off_l)
    assert coder.rans_encode(sym, idx, gc_tables) == want
    d_ref = ref["ans"].RansDecoder()
    d_ref.set_stream(want)
    d = coder.RansDecoder()
    d.set_stream(want)
    cut = n // 3
    for a, b in ((0, cut), (cut, n)):
        assert d.decode_stream(idx[a:b], gc_tables).tolist() == d_ref.decode_stream(idx[a:b].tolist(), cdf_l, sz_l, off_l)
    for trial in range(20):
        m = rng.randint(2, 40)
        p = rng.rand(m).astype(np.float32) ** 4
        p /= p.sum()
        assert coder.pmf_to_quantized_cdf(p).tolist() == ref["_CXX"].pmf_to_quantized_cdf([float(v) for v in p], 16)


@pytest.mark.parametrize("seed", [0, 1])
def test_lane_edge_tables_against_reference_cpp(seed):
    """The table family of tests/test_gpu_coder.py::test_rows_around_the_lane_count (rows of 2 ... 70 slots, frequency-1
    symbols, escapes on both sides) through the reference's own C++ coder: pins the oracle on exactly the inputs the GPU
    decoder's lane-count edges are tested with."""
    from coder_cases import lane_edge_symbols, lane_edge_tables

    ref = coder.load_reference_coder()
    if ref is None:
        pytest.skip("oracle/_ref not built (make -C oracle ref needs /root/reference)")
    cdf, sizes, offsets, rng = lane_edge_tables(seed)
    ot = coder.Tables(cdf, sizes, offsets)
    cdf_l, sz_l, off_l = cdf.tolist(), sizes.tolist(), offsets.tolist()
    for n in (1, 63, 64, 65, 4000):
        idx, sym = lane_edge_symbols(rng, n, sizes, offsets)
        if n < 2:  # (the reference's flush() under-allocates its output for fewer than two symbols: rans_interface.cpp:171)
            continue
        want = ref["ans"].RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdf_l, sz_l, off_l)
        assert coder.rans_encode(sym, idx, ot) == want
        d_ref = ref["ans"].RansDecoder()
        d_ref.set_stream(want)
        assert d_ref.decode_stream(idx.tolist(), cdf_l, sz_l, off_l) == sym.tolist()
        assert coder.rans_decode(want, idx, ot).tolist() == sym.tolist()


def test_coarse_edge_tables_against_reference_cpp():
    """The table family of tests/test_gpu_coder.py::test_rows_around_the_coarse_level (rows of 128 ... 4033 slots next to narrow
    ones, every slot, escapes on both sides) through the reference's own C++ coder: the oracle is pinned on exactly the inputs
    the GPU decoder's coarse first level is tested with."""
    from coder_cases import COARSE_EDGE_SLOTS, lane_edge_symbols, lane_edge_tables

    ref = coder.load_reference_coder()
    if ref is None:
        pytest.skip("oracle/_ref not built (make -C oracle ref needs /root/reference)")
    cdf, sizes, offsets, rng = lane_edge_tables(0, COARSE_EDGE_SLOTS)
    ot = coder.Tables(cdf, sizes, offsets)
    cdf_l, sz_l, off_l = cdf.tolist(), sizes.tolist(), offsets.tolist()
    coarse = [r for r, n in enumerate(COARSE_EDGE_SLOTS) if 128 < n <= 4032]
    for n, rows in ((65, None), (3000, coarse), (3000, None)):
        idx, sym = lane_edge_symbols(rng, n, sizes, offsets, rows=rows)
        want = ref["ans"].RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdf_l, sz_l, off_l)
        assert coder.rans_encode(sym, idx, ot) == want
        d_ref = ref["ans"].RansDecoder()
        d_ref.set_stream(want)
        assert d_ref.decode_stream(idx.tolist(), cdf_l, sz_l, off_l) == sym.tolist()
        assert coder.rans_decode(want, idx, ot).tolist() == sym.tolist()
