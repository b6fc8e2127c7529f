"""Input families shared by the CPU (oracle vs the reference's C++) and GPU (HIP coder vs oracle) coder tests."""
import numpy as np


def lane_edge_tables(seed, slots=(2, 3, 5, 17, 31, 32, 33, 62, 63, 64, 65, 66, 70)):
    """CDF rows of 2 ... 70 slots with random frequencies (many of them 1): the GPU decoder's first level holds 64 slots of a
    row in the 64 lanes, so 63 / 64 / 65 slots are its edges.  Returns (cdf [rows, stride], sizes, offsets, rng)."""
    rng = np.random.RandomState(100 + seed)
    slots = list(slots)
    stride = max(slots) + 1
    cdf = np.zeros((len(slots), stride), np.int32)
    for r, n in enumerate(slots):
        f = np.ones(n, np.int64)
        f += rng.multinomial(65536 - n, rng.dirichlet(np.full(n, 0.3)))
        cdf[r, : n + 1] = np.concatenate([[0], np.cumsum(f)])
        assert cdf[r, n] == 65536
    sizes = np.array([n + 1 for n in slots], np.int32)
    offsets = np.array([-(n // 2) for n in slots], np.int32)
    return cdf, sizes, offsets, rng


# rows around the decoder's coarse first level (129 ... 4032 slots: blocks of ceil(slots / 64) symbols), next to narrow rows and
# to wide rows that stay with the bucket table (65 ... 128 and > 4032 slots)
COARSE_EDGE_SLOTS = (5, 40, 64, 100, 128, 129, 130, 191, 192, 193, 1000, 4031, 4032, 4033)


def lane_edge_symbols(rng, n, sizes, offsets, rows=None):
    """n (index, symbol) pairs over those rows (or over `rows`, a sequence of row indices to draw from): every table slot, the
    first and the last one over-represented, and one in ten an escape on either side of the table."""
    idx = (rng.randint(0, len(sizes), n) if rows is None else np.asarray(rows)[rng.randint(0, len(rows), n)]).astype(np.int32)
    v = (rng.rand(n) * (sizes[idx] - 2)).astype(np.int64)      # a table slot (escape slot excluded) ...
    edge = rng.rand(n)
    v = np.where(edge < 0.15, 0, np.where(edge < 0.3, sizes[idx] - 3, v))
    sym = v + offsets[idx]
    esc = rng.rand(n) < 0.1                                     # ... or an escape on either side
    far = (2.0 ** rng.uniform(0, 20, n)).astype(np.int64)
    sym = np.where(esc, np.where(rng.rand(n) < 0.5, offsets[idx] - far, offsets[idx] + sizes[idx] - 2 + far), sym)
    return idx, sym.astype(np.int32)
