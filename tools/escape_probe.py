import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import rgbd_amd
from rgbd_amd import ans
from rgbd_amd.entropy_models import GaussianConditional, get_scale_table
from oracle import coder
gc = GaussianConditional(); gc.update_scale_table(get_scale_table(), force=True)
cdf, sizes, offsets = gc.numpy_tables()
t = ans.Tables(cdf, sizes, offsets); ot = coder.Tables(cdf, sizes, offsets)
scales = np.exp(np.linspace(np.log(0.11), np.log(256), 64))
import sys as _s
CASES = [tuple(int(v) if i != 1 else float(v) for i, v in enumerate(a.split(","))) for a in _s.argv[1:]] or [(300, 0.2, 6)]
for n, rate, maxbits in CASES:
    rng = np.random.RandomState(n + maxbits)
    idx = rng.randint(0, 64, n).astype(np.int32)
    sym = np.rint(rng.standard_normal(n) * scales[idx]).astype(np.int64)
    esc = rng.rand(n) < rate
    mag = (2.0 ** rng.uniform(0, maxbits, int(esc.sum()))).astype(np.int64)
    sym[esc] = np.where(rng.rand(int(esc.sum())) < 0.5, mag, -mag)
    sym = sym.astype(np.int32)
    print("case", n, rate, maxbits, "encoding", flush=True)
    s = ans._encode(t, sym, idx)
    print("  encoded", len(s), "equal to oracle:", s == coder.rans_encode(sym, idx, ot), flush=True)
    d = ans.RansDecoder(); d.set_stream(coder.rans_encode(sym, idx, ot))
    print("  decoding", flush=True)
    out = np.asarray(d.decode_stream(idx, cdf, sizes, offsets), np.int32)
    print("  decoded equal:", np.array_equal(out, sym), flush=True)
