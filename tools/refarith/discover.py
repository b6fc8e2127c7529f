"""Measure, on the reference machine, the shape-dependent part of the CPU arithmetic the reference's float path ends in, and
write it to learning-based-rgb-d-image-compression_amd/refarith_tables.json (committed DATA; the product reads it, never torch).

CONTAINER-ONLY TOOL (needs torch CPU = the library stack the reference runs on; it is a black-box probe of installed
third-party libraries -- oneDNN, MKL -- and reads nothing under /root/reference).

How a summation structure is read off a black box: all products are made 0 / 1 except one of 2^25 -- every 1 that is added
to the running sum AFTER the big value is absorbed (2^25 + 1 == 2^25 in fp32), every 1 added before it (or in another
accumulator) survives.  The output then tells how many terms follow the probed term inside its own fma chain; walking the
probe over the input channels gives the chains ("blocks") and their order.  oracle/cpu_arith.c executes the structures found,
and tests/test_oracle_arith.py checks them against torch bit for bit on random data.

usage: python tools/refarith/discover.py                       (all golden / bench shapes; ~10 min)
       python tools/refarith/discover.py --add united:H:W:B ...   (measure the layer shapes of further image sizes -- kind is
                                                                   united / single / r2d -- and MERGE them into the committed
                                                                   tables: what a user of another image size runs once)
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "learning-based-rgb-d-image-compression_amd", "refarith_tables.json")
BIG = 2.0 ** 25


def uses_mkldnn(xshape, wshape):
    """torch's ConvParams::use_mkldnn for fp32 CPU tensors with more than one thread: the small-tensor exception."""
    n = xshape[0]
    k_big = wshape[-1] > 3 and wshape[-2] > 3
    return not (n == 1 and not k_big and int(np.prod(xshape)) <= 20480)


def probe_1x1_blocks(cin, cout, h, w, b):
    """reduce blocks of the 1x1 convolution: channel c's probe output = cin - 1 - (#terms after c in its chain)"""
    res = np.zeros(cin)
    co = max(cout, 16)
    for g0 in range(0, cin, co):
        wt = torch.ones(co, cin, 1, 1)
        n = min(co, cin - g0)
        for o in range(n):
            wt[o, g0 + o, 0, 0] = BIG
        y = F.conv2d(torch.ones(b, cin, h, w), wt, torch.zeros(co))
        res[g0:g0 + n] = cin - 1 - (y[0, :n, h // 2, w // 2] - BIG).numpy()
    blocks, start = [], 0
    for i in range(1, cin):
        if res[i] > res[i - 1]:
            blocks.append(i - start)
            start = i
    blocks.append(cin - start)
    return blocks


def probe_im2col_kblocks(cin, cout, k, h, w, pad, stride=1):
    """K blocks of the small-tensor path (im2col + sgemm), k index = c*k*k + ky*k + kx, probed at a central output pixel"""
    OH, OW = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    oy, ox = OH // 2, OW // 2
    terms = [(c, ky, kx) for c in range(cin) for ky in range(k) for kx in range(k)
             if 0 <= oy * stride - pad + ky < h and 0 <= ox * stride - pad + kx < w]
    nv = len(terms)
    res = {}
    co = max(cout, 16)
    for g0 in range(0, nv, co):
        grp = terms[g0:g0 + co]
        wt = torch.ones(co, cin, k, k)
        for o, (c, ky, kx) in enumerate(grp):
            wt[o, c, ky, kx] = BIG
        y = F.conv2d(torch.ones(1, cin, h, w), wt, torch.zeros(co), stride=stride, padding=pad)
        for o, t in enumerate(grp):
            res[t] = nv - 1 - float(y[0, o, oy, ox] - BIG)
    # block boundaries in k (the probed pixel may miss border taps: boundaries are reported in full-k units)
    seq = [(c * k * k + ky * k + kx, res[(c, ky, kx)]) for (c, ky, kx) in terms]
    bounds = [0]
    for i in range(1, len(seq)):
        if seq[i][1] > seq[i - 1][1] + 0.5:
            # the block starts somewhere in (last term of the old block, first term of the new one]: border taps the probed
            # pixel does not have leave a gap -- take the channel boundary inside it when there is one
            lo, hi = seq[i - 1][0] + 1, seq[i][0]
            cands = [b for b in range(lo, hi + 1) if b % (k * k) == 0]
            bounds.append(cands[-1] if cands else hi)
    bounds.append(cin * k * k)
    return [bounds[i + 1] - bounds[i] for i in range(len(bounds) - 1)]


def classify_linear_rows(K, J, samples=6):
    """nn.Linear(K -> J, bias=False) applied to ONE vector: per output row, which of the dot-product orders of
    oracle/cpu_arith.c (0 main, 1 / 2 remainder rows with one / two accumulators) reproduces torch on random data"""
    import ctypes
    from oracle import cpu_arith as ca

    L = ca.lib()
    L.orc_dot_main.restype = ctypes.c_float
    L.orc_dot_rem.restype = ctypes.c_float
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    g = torch.Generator().manual_seed(K * 7919 + J)
    W = (torch.randn(J, K, generator=g) / K ** 0.5).contiguous()
    xs = [torch.randn(1, K, generator=g) for _ in range(samples)]
    refs = [F.linear(x, W)[0].numpy() for x in xs]
    Wn = W.numpy()
    out = []
    for j in range(J):
        fit = None
        for cls in (0, 2, 1):
            ok = True
            for x, r in zip(xs, refs):
                xn = x[0].numpy()
                v = L.orc_dot_main(P(Wn[j]), P(xn), K) if cls == 0 else L.orc_dot_rem(P(Wn[j]), P(xn), K, cls)
                if np.float32(v) != r[j]:
                    ok = False
                    break
            if ok:
                fit = cls
                break
        if fit is None:
            raise RuntimeError(f"Linear {K}->{J}: row {j} fits none of the known accumulation orders")
        out.append(fit)
    return out


def _dc_ranks_at(cin, cout, k, h, w, oy, ox, b=1):
    """stride-2 transposed conv: for every tap that reaches output pixel (oy, ox), its position from the END of its fma chain
    (in units of taps = cin terms)"""
    pad = k // 2
    taps = [(ky, kx) for ky in range(k) for kx in range(k) if (oy + pad - ky) % 2 == 0 and (ox + pad - kx) % 2 == 0
            and 0 <= (oy + pad - ky) // 2 < h and 0 <= (ox + pad - kx) // 2 < w]
    nv = len(taps) * cin
    res = {}
    for g0 in range(0, len(taps), cout):
        grp = taps[g0:g0 + cout]
        wt = torch.ones(cin, cout, k, k)
        for o, (ky, kx) in enumerate(grp):
            wt[0, o, ky, kx] = BIG
        y = F.conv_transpose2d(torch.ones(b, cin, h, w), wt, None, stride=2, padding=pad, output_padding=1)
        for o, t in enumerate(grp):
            res[t] = int(round((nv - 1 - float(y[0, o, oy, ox] - BIG) + 1) / cin))
    return res


def _dc_chain_order(cin, cout, k, h, w, oy, ox, chains, b=1):
    """the order in which the chain sums are added: the LAST chain is the one whose value survives a +BIG / -BIG pair placed
    in any two other chains; remove it and repeat"""
    import itertools

    pad = k // 2
    rem, order = list(range(len(chains))), []
    x = torch.ones(b, cin, h, w)
    while len(rem) > 2:
        found = None
        for L in rem:
            pairs = list(itertools.combinations([c for c in rem if c != L], 2))
            ok = True
            for g0 in range(0, len(pairs), cout):
                grp = pairs[g0:g0 + cout]
                wt = torch.zeros(cin, cout, k, k)
                for o, (i, j) in enumerate(grp):
                    wt[0, o, chains[i][0][0], chains[i][0][1]] = BIG
                    wt[0, o, chains[j][0][0], chains[j][0][1]] = -BIG
                    wt[0, o, chains[L][0][0], chains[L][0][1]] = 1.0
                y = F.conv_transpose2d(x, wt, None, stride=2, padding=pad, output_padding=1)
                if not all(float(y[0, o, oy, ox]) == 1.0 for o in range(len(grp))):
                    ok = False
                    break
            if ok:
                found = L
                break
        if found is None:
            raise RuntimeError("deconv: no consistent chain order")
        order.append(found)
        rem.remove(found)
    return rem + order[::-1]


def _dc_recipe_at(cin, cout, k, h, w, oy, ox, b=1):
    r = _dc_ranks_at(cin, cout, k, h, w, oy, ox, b)
    taps = sorted(r)
    kys = sorted(set(t[0] for t in taps))
    kxs = sorted(set(t[1] for t in taps))
    if all(v == 1 for v in r.values()):
        chains = [[t] for t in taps]
    else:  # chains are sets of whole kx columns walked ky -> kx: the last ky row's ranks tell which columns share one
        last, groups, cur = kys[-1], [], []
        for kx in kxs:
            cur.append(kx)
            if r[(last, kx)] == 1:
                groups.append(cur)
                cur = []
        if cur:
            raise RuntimeError(f"deconv: unexpected chain structure {r}")
        chains = [[(ky, kx) for ky in kys for kx in g if (ky, kx) in r] for g in groups]
        for ch in chains:
            for i, t in enumerate(ch):
                if r[t] != len(ch) - i:
                    raise RuntimeError(f"deconv: unexpected chain structure {r}")
    if len(chains) > 2:
        chains = [chains[i] for i in _dc_chain_order(cin, cout, k, h, w, oy, ox, chains, b)]
    return chains


def probe_deconv_s2(cin, cout, k, h, w, b=1):
    """recipe per (py, px, j = ox / 2): chains of taps in addition order, flattened as
    [n, ky, kx, fresh, ky, kx, fresh, ...] for the 4 * w (phase, column) pairs in order; checked against torch on random data"""
    import ctypes
    from oracle import cpu_arith as ca

    rec = {}
    for py in (0, 1):
        for px in (0, 1):
            rows = [2 * (h // 2) + py] if h > 3 else [2 * i + py for i in range(h)]
            for j in range(w):
                ox = 2 * j + px
                per_row = [_dc_recipe_at(cin, cout, k, h, w, oy, ox, b) for oy in rows]
                if len(per_row) == 1:
                    rec[(py, px, j)] = per_row[0]
                    continue
                taps = sorted(set(t for ch in per_row for c in ch for t in c))
                if all(all(len(c) == 1 for c in ch) for ch in per_row):
                    rec[(py, px, j)] = [[t] for t in sorted(taps, key=lambda t: (t[1], t[0]))]
                else:
                    groups = []
                    for ch in per_row:
                        for c in ch:
                            g = tuple(sorted(set(t[1] for t in c)))
                            if g not in groups:
                                groups.append(g)
                    groups.sort()
                    rec[(py, px, j)] = [[(ky, kx) for ky in sorted(set(t[0] for t in taps)) for kx in g if (ky, kx) in taps]
                                        for g in groups]
    flat, off = [], []
    for py in (0, 1):
        for px in (0, 1):
            for j in range(w):
                off.append(len(flat))
                f = [(ky, kx, 1 if i == 0 else 0) for c in rec[(py, px, j)] for i, (ky, kx) in enumerate(c)]
                flat += [len(f)] + [v for t in f for v in t]
    # verify on random data
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    g = torch.Generator().manual_seed(cin + h * 131 + w)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, k, k, generator=g) / (cin * 6) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv_transpose2d(x, wt, bias, stride=2, padding=k // 2, output_padding=1).numpy()
    y = np.empty_like(ref)
    offa, desca = np.array(off, np.int32), np.array(flat, np.int32)
    ca.lib().orc_deconv_s2(P(np.ascontiguousarray(x.numpy())), b, cin, h, w, P(np.ascontiguousarray(wt.numpy())), cout, k,
                           P(bias.numpy()), P(offa), P(desca), P(y))
    if not np.array_equal(y, ref):
        raise RuntimeError(f"deconv {cin}->{cout} {h}x{w}: the measured recipe does not reproduce torch")
    return flat


def collect_shapes(kind, H, W, B):
    """every conv call of one compress + decompress (and forward) of the oracle at this input shape"""
    from oracle import elic_oracle as eo
    from rgbd_amd import synth

    calls = []
    oc, od = eo._conv, eo._deconv

    def lc(sd, name, x, stride=1, pad=None):
        w = sd[name + ".weight"]
        p = w.shape[-1] // 2 if pad is None else pad
        calls.append(("conv", tuple(x.shape), tuple(w.shape), stride, p))
        return oc(sd, name, x, stride, pad)

    def ld(sd, name, x, stride):
        w = sd[name + ".weight"]
        calls.append(("deconv:" + name.split(".")[0], tuple(x.shape), tuple(w.shape), stride, w.shape[-1] // 2))
        return od(sd, name, x, stride)

    eo._conv, eo._deconv = lc, ld
    try:
        if kind == "united":
            c = eo.OracleCodec(synth.synthetic_state_dict(0))
            c.update()
            r, d = synth.synthetic_batch(B, H, W, config_id=2)
            rp, dp = eo.pad_replicate0(torch.from_numpy(r)), eo.pad_replicate0(torch.from_numpy(d))
            out = c.compress(rp, dp)
            c.decompress(out["r_strings"], out["d_strings"], out["shape"])
        elif kind == "r2d":
            c = eo.oracle_r2d(synth.synthetic_state_dict(0, model="ELIC_united_R2D"))
            c.update()
            r, d = synth.synthetic_batch(B, H, W, config_id=2)
            rp, dp = eo.pad_replicate0(torch.from_numpy(r)), eo.pad_replicate0(torch.from_numpy(d))
            out = c.compress(rp, dp)
            c.decompress(out["r_strings"], out["d_strings"], out["shape"])
        elif kind == "single":
            c = eo.OracleCodecSingle(synth.synthetic_state_dict(0, model="ELIC"))
            c.update()
            r, _ = synth.synthetic_batch(B, H, W, config_id=1)
            rp = eo.pad_replicate0(torch.from_numpy(r))
            out = c.compress(rp)
            c.decompress(out["strings"], out["shape"])
        else:
            raise ValueError(kind)
    finally:
        eo._conv, eo._deconv = oc, od
    return sorted(set(calls))


def main():
    torch.set_num_threads(8)
    jobs = [("united", 128, 192, 1), ("united", 128, 128, 2), ("united", 128, 192, 2), ("united", 256, 256, 1),
            ("united", 480, 640, 1), ("single", 256, 256, 1), ("r2d", 128, 192, 1)]
    if "--quick" in sys.argv:
        jobs = jobs[:1]
    tab = {"meta": {"torch": torch.__version__, "threads": torch.get_num_threads(),
                    "note": "measured by tools/refarith/discover.py on the machine that produced tests/golden/"},
           "conv1x1": [], "im2col": [], "deconv_s2": []}
    seen1, seen2, seen3 = set(), set(), set()
    old = None
    if "--add" in sys.argv:  # only the named image sizes, merged into what is committed
        jobs = []
        for a in sys.argv[sys.argv.index("--add") + 1:]:
            if a.startswith("--"):
                break
            kind, H, W, B = a.split(":")
            jobs.append((kind, int(H), int(W), int(B)))
        with open(OUT) as f:
            old = json.load(f)
        if old["meta"].get("threads") != torch.get_num_threads():
            raise SystemExit(f"the committed tables were measured with {old['meta'].get('threads')} threads, this process has "
                             f"{torch.get_num_threads()}: the CPU kernels block differently, do not mix")
        tab["meta"] = old["meta"]
        tab["conv1x1"], tab["im2col"], tab["deconv_s2"] = list(old["conv1x1"]), list(old["im2col"]), list(old.get("deconv_s2", []))
        seen1 = {tuple(e[:5]) for e in tab["conv1x1"]}
        seen2 = {tuple(e[:7]) for e in tab["im2col"]}
        seen3 = {tuple(e[:6]) for e in tab["deconv_s2"]}
    for kind, H, W, B in jobs:
        try:
            shapes = collect_shapes(kind, H, W, B)
        except Exception as e:  # a model kind the synthetic-weight generator does not know: skip, say so
            print("skip", kind, H, W, B, repr(e)[:120])
            continue
        for (op, xs, ws, stride, pad) in shapes:
            if op == "deconv:h_s" and stride == 2:  # the hyper-synthesis stages feed every entropy parameter
                key = (xs[1], ws[1], ws[2], xs[2], xs[3], xs[0])
                if key not in seen3:
                    seen3.add(key)
                    tab["deconv_s2"].append(list(key) + [probe_deconv_s2(*key)])
                continue
            if op != "conv":
                continue
            n, cin, h, w = xs
            cout, _, k, _ = ws
            if uses_mkldnn(xs, ws):
                if k == 1:
                    key = (cin, cout, h, w, n)
                    if key not in seen1:
                        seen1.add(key)
                        tab["conv1x1"].append([cin, cout, h, w, n, probe_1x1_blocks(cin, cout, h, w, n)])
            else:
                key = (cin, cout, k, h, w, stride, pad)
                if key not in seen2:
                    seen2.add(key)
                    tab["im2col"].append([cin, cout, k, h, w, stride, pad, probe_im2col_kblocks(cin, cout, k, h, w, pad, stride)])
        print(kind, H, W, B, "->", len(tab["conv1x1"]), "1x1 entries,", len(tab["im2col"]), "im2col entries", flush=True)
    tab["conv1x1"].sort()
    tab["im2col"].sort()
    # SE_Block Linear layers (batch 1): row classes per (K, J)
    from rgbd_amd import synth
    lin = set()
    for model in (None, "ELIC", "ELIC_united_R2D"):
        sd = synth.synthetic_state_dict(0) if model is None else synth.synthetic_state_dict(0, model=model)
        for name, t in sd.items():
            if ".se.fc." in name and name.endswith(".weight"):
                lin.add((int(t.shape[1]), int(t.shape[0])))
    tab["linear"] = []
    for K, J in sorted(lin):
        cls = classify_linear_rows(K, J)
        # run-length encoded: [class, count, class, count, ...]
        rle = []
        for c in cls:
            if rle and rle[-2] == c:
                rle[-1] += 1
            else:
                rle += [c, 1]
        tab["linear"].append([K, J, rle])
    if old is not None:  # (--add: the SE shapes do not depend on the image size)
        tab["linear"] = old["linear"]
    print(len(tab["linear"]), "linear entries", flush=True)
    with open(OUT, "w") as f:
        f.write("{\n")
        f.write('"meta": ' + json.dumps(tab["meta"]) + ",\n")
        tab["deconv_s2"].sort()
        sects = ("conv1x1", "im2col", "linear", "deconv_s2")
        for sect in sects:
            f.write(f'"{sect}": [\n' + ",\n".join(json.dumps(e) for e in tab[sect]) + "\n]" + ("," if sect != sects[-1] else "") + "\n")
        f.write("}\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
