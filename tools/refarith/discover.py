"""Measure, on the reference machine, the shape-dependent part of the CPU arithmetic the reference's float path ends in, and
write it to learning-based-rgb-d-image-compression_amd/refarith_tables.json (committed DATA; the product reads it, never torch).

CONTAINER-ONLY TOOL (needs torch CPU = the library stack the reference runs on; it is a black-box probe of installed
third-party libraries -- oneDNN, MKL -- and reads nothing under /root/reference).

How a summation structure is read off a black box: all products are made 0 / 1 except one of 2^25 -- every 1 that is added
to the running sum AFTER the big value is absorbed (2^25 + 1 == 2^25 in fp32), every 1 added before it (or in another
accumulator) survives.  The output then tells how many terms follow the probed term inside its own fma chain; walking the
probe over the input channels gives the chains ("blocks") and their order.  oracle/cpu_arith.c executes the structures found,
and tests/test_oracle_arith.py checks them against torch bit for bit on random data.

usage: python tools/refarith/discover.py            (all golden / bench shapes; ~10 min)
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
        calls.append(("deconv", tuple(x.shape), tuple(w.shape), stride, w.shape[-1] // 2))
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
    finally:
        eo._conv, eo._deconv = oc, od
    return sorted(set(calls))


def main():
    torch.set_num_threads(8)
    jobs = [("united", 128, 192, 1), ("united", 128, 128, 2), ("united", 128, 192, 2), ("united", 256, 256, 1),
            ("united", 480, 640, 1)]
    if "--quick" in sys.argv:
        jobs = jobs[:1]
    tab = {"meta": {"torch": torch.__version__, "threads": torch.get_num_threads(),
                    "note": "measured by tools/refarith/discover.py on the machine that produced tests/golden/"},
           "conv1x1": [], "im2col": []}
    seen1, seen2 = set(), set()
    for kind, H, W, B in jobs:
        try:
            shapes = collect_shapes(kind, H, W, B)
        except Exception as e:  # a model kind the synthetic-weight generator does not know: skip, say so
            print("skip", kind, H, W, B, repr(e)[:120])
            continue
        for (op, xs, ws, stride, pad) in shapes:
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
    with open(OUT, "w") as f:
        f.write("{\n")
        f.write('"meta": ' + json.dumps(tab["meta"]) + ",\n")
        for sect in ("conv1x1", "im2col"):
            f.write(f'"{sect}": [\n' + ",\n".join(json.dumps(e) for e in tab[sect]) + "\n]" + ("," if sect == "conv1x1" else "") + "\n")
        f.write("}\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
