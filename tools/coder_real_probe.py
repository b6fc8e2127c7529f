#!/usr/bin/env python3
"""The symbols of a real compress() (one 512x640 pair, modality 0, stream order) through the stand-alone coder ABI in ONE
launch each: ns/symbol of the serial chain on the model's own symbol statistics, next to what the 20 part launches of a
decompress() take in place.  Usage: rocprofv3 --kernel-trace --output-format csv -d out -o p -- python3 tools/coder_real_probe.py
then: python3 tools/coder_real_probe.py --report out"""
import csv
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

RECIPES = ("stress", "trained_like")
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    tr = glob.glob(f"{sys.argv[2]}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])  # noqa: E731
    enc = [dur(r) for r in rows if "rans_encode" in r["Kernel_Name"]]
    dec = [dur(r) for r in rows if "rans_decode" in r["Kernel_Name"]]
    meta = [l.split() for l in open(os.path.join(sys.argv[2], "real_probe_meta.txt"))]
    # per recipe: two rounds of compress() (z rgb, z depth, y of both modalities = 3 encode launches) + decompress() (2 z +
    # 20 y part launches), then the stand-alone pair over modality 0's symbols; the second round is the one reported
    ei = di = 0
    for name, n, esc, nbytes in meta:
        n = int(n)
        in_enc, in_dec = enc[ei + 5], sum(dec[di + 24:di + 44])
        solo_enc, solo_dec = enc[ei + 6], dec[di + 44]
        ei, di = ei + 7, di + 45
        print(f"{name:13s} {n} symbols per modality, {float(esc) * 100:.1f} % escapes, {float(nbytes) * 8 / n:.2f} bits/symbol | in place: "
              f"encode {in_enc / n:6.1f} ns/symbol (two streams in parallel), decode {in_dec / (2 * n):6.1f} ns/symbol (20 launches, "
              f"rgb and depth parts back to back = {in_dec / 1e6:.1f} ms) | one launch, one modality: encode {solo_enc / n:6.1f} decode {solo_dec / n:6.1f}")
    sys.exit(0)

import torch  # noqa: E402

import rgbd_amd  # noqa: E402
from rgbd_amd import ELIC_united, ans, synth  # noqa: E402
from rgbd_amd.entropy_models import GaussianConditional, get_scale_table  # noqa: E402

gc = GaussianConditional()
gc.update_scale_table(get_scale_table(), force=True)
cdf, sizes, offsets = gc.numpy_tables()
t = ans.Tables(cdf, sizes, offsets)
out_dir = os.environ.get("REAL_PROBE_OUT", ".")
meta = []
for recipe in RECIPES:
    sd = synth.synthetic_state_dict(0, recipe=recipe)
    net = ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
    net.load_state_dict(sd)
    net.update(force=True)
    net = net.to("cuda")
    r, d = synth.synthetic_batch(1, 512, 640, config_id=3)
    rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
    for _ in range(2):  # the second pair of calls is the one the report reads
        out = net.compress(rgb, depth)
        net.decompress(out["r_strings"], out["d_strings"], out["shape"])
    torch.cuda.synchronize()
    sym, idx = net.debug_symbols(0)
    v = sym - offsets[idx]
    esc = float(np.mean((v < 0) | (v >= sizes[idx] - 2)))
    s = ans._encode(t, sym, idx)
    dec = ans.RansDecoder()
    dec.set_stream(s)
    got = np.asarray(dec.decode_stream(idx, cdf, sizes, offsets), dtype=np.int32)
    assert np.array_equal(got, sym), recipe
    meta.append((recipe, len(sym), esc, len(s)))
    print(recipe, len(sym), "symbols", f"{esc * 100:.1f} % escapes", len(s), "bytes", flush=True)
    del net
with open(os.path.join(out_dir, "real_probe_meta.txt"), "w") as f:
    for m in meta:
        f.write(" ".join(str(x) for x in m) + "\n")
