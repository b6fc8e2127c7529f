#!/usr/bin/env python3
"""In-situ tile tuner: every conv layer of a REAL compress() + decompress() is timed under every candidate tile / staging form --
inside the model, in launch order, with the operands the previous layer left in the caches -- instead of as a stand-alone
kernel (tools/tune_tiles.py), whose winners did not carry over to the 16-image calls (profiles/r05_call_batch_sweep.txt).

One pass r of the codec runs EVERY layer shape under its r-th candidate (rgbd_debug_tile_override); the profile (HIP events
around every conv launch of one engine instance, rgbd_elic_set_profile(m, 2): names carry the shape key) gives each shape's
time under that candidate, so ~25 passes cover 25 candidates for all shapes at once.  Candidates are filtered first by one
kernel-only launch (rgbd_conv_bench) so that a pass never meets a form a shape cannot take.  Tile choice never changes a result.

    python tools/tune_insitu.py [--write] [--latency] [--model STF_united] B H W
           (default 16 512 640, ELIC_united; writes gpurun_out/insitu_*.h, --write: csrc/; --latency: the tables a lone engine
            instance uses, tile_table[_blk].h)
"""
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rgbd_amd  # noqa: E402
from rgbd_amd import ELIC_united, synth  # noqa: E402
from rgbd_amd._lib import lib  # noqa: E402

MODEL = sys.argv[sys.argv.index("--model") + 1] if "--model" in sys.argv else "ELIC_united"
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--model"]
B, H, W = (int(v) for v in args[:3]) if len(args) >= 3 else (16, 512, 640)
WRITE = "--write" in sys.argv
LATENCY = "--latency" in sys.argv
REPS = 3
L = lib()
net = rgbd_amd.modelZoo[MODEL](config=rgbd_amd.model_config(), channel=4).eval()
net.load_state_dict(synth.synthetic_state_dict(0, model=MODEL) if MODEL != "ELIC_united" else synth.synthetic_state_dict(0))
net.update(force=True)
net = net.to("cuda")
net.per_image_streams = True
net.set_tile_mode("latency" if LATENCY else "throughput")
r, d = synth.synthetic_batch(B, H, W, config_id=2)
rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()


def roundtrip():
    out = net.compress(rgb, depth)
    net.decompress(out["r_strings"], out["d_strings"], out["shape"])


def profile_pass():
    """-> {key: ms} (non-fused launches only), summed over the call, best of REPS"""
    best = {}
    for _ in range(REPS):
        L.rgbd_elic_set_profile(net._h, 2)
        roundtrip()
        path = b"/tmp/insitu_layers.csv"
        L.rgbd_elic_profile_dump(net._h, path)
        L.rgbd_elic_set_profile(net._h, 0)
        cur = collections.defaultdict(float)
        for ln in open(path.decode()).read().strip().split("\n")[1:]:
            name = ln.split(",", 1)[0] if "|" not in ln else None
            if name is not None:
                continue
            nm, key, rest = ln.split("|")
            fused, cnt, ms = rest.split(",")[:3]
            if fused == "0":
                cur[key] += float(ms)
        for k, v in cur.items():
            best[k] = min(best.get(k, 1e30), v)
    return best


L.rgbd_elic_profile_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
for _ in range(2):
    roundtrip()
base = profile_pass()
keys = sorted(base, key=lambda k: -base[k])
print(f"{len(keys)} layer shapes, {sum(base.values()):.2f} ms of non-fused conv per call (B = {B})", flush=True)

TILES = ([(2, m, 8) for m in (2, 1)] + [(2, m, n) for n in (4, 2, 1) for m in (5, 4, 3, 2, 1)] + [(1, m, n) for n in (4, 2, 1) for m in (3, 2, 1)])
MODES = ((16, 1), (16, 2), (16, 3), (16, 0), (64, 0), (16, 4), (16, 5))


def valid(key, cand):
    N, Hh, Ww, cin, cout, ntaps, stride, nphase, splitk = key
    k = int(round(ntaps ** 0.5))
    L.rgbd_debug_force_blocked(1 if nphase >= 100 else 0)
    ph = nphase % 100
    L.rgbd_debug_force_ckbd(ph // 10)
    ph %= 10
    L.rgbd_debug_force_splitk(splitk)
    L.rgbd_debug_force_tile(",".join(str(v) for v in cand).encode())
    ms = ctypes.c_float(0)
    rc = L.rgbd_conv_bench(N, cin, Hh, Ww, cout, k, stride, k // 2, 1 if ph > 1 else 0, 0, 1, ctypes.byref(ms))
    return rc == 0


cands = {}
for ks in keys:
    if base[ks] < 0.02:  # (< 20 us per call: nothing to win)
        continue
    key = tuple(int(v) for v in ks.split(","))
    ok = [(wm, mt, nt, kc, dm) for wm, mt, nt in TILES for kc, dm in MODES if valid(key, (wm, mt, nt, kc, dm))]
    cands[ks] = ok
L.rgbd_debug_force_tile(b"")
L.rgbd_debug_force_splitk(0)
L.rgbd_debug_force_ckbd(0)
L.rgbd_debug_force_blocked(0)
R = max(len(v) for v in cands.values())
print(f"{len(cands)} shapes tuned, up to {R} candidates each", flush=True)
times = {ks: {} for ks in cands}
for rnd in range(R):
    lines = [ks + "," + ",".join(str(v) for v in c[rnd]) for ks, c in cands.items() if rnd < len(c)]
    assert L.rgbd_debug_tile_override("\n".join(lines).encode()) == 0
    got = profile_pass()
    for ks, c in cands.items():
        if rnd < len(c) and ks in got:
            times[ks][c[rnd]] = got[ks]
    print(f"pass {rnd + 1}/{R}", flush=True)
L.rgbd_debug_tile_override(b"")
# Every tuned shape gets an entry -- its fastest candidate -- not only the shapes whose winner beats the current choice: the
# table lookup treats a (map, batch) it has entries for as MEASURED and sends every other layer of that map to the cost
# model instead of to a neighbouring batch size's entry, so a table of winners only would take the neighbour entries away from
# all the layers it does not list (what made the first attempts slower end to end, profiles/r05_call_batch_sweep.txt).
best = {ks: min(t.items(), key=lambda kv: kv[1]) for ks, t in times.items() if t}
assert L.rgbd_debug_tile_override("\n".join(ks + "," + ",".join(str(v) for v in c) for ks, (c, _) in best.items()).encode()) == 0
conf = profile_pass()
L.rgbd_debug_tile_override(b"")
base2 = profile_pass()
keep = {ks: c for ks, (c, _) in best.items()}
tot_b = sum(min(base[k], base2.get(k, 1e30)) for k in base)
tot_c = sum(conf.get(k, base[k]) for k in base)
print(f"non-fused conv per call: {tot_b:.2f} ms (current choice) -> {tot_c:.2f} ms with the fastest candidate of each of {len(keep)} shapes")
for ks in sorted(keep, key=lambda k: -base[k]):
    print(f"  {ks}: {min(base[ks], base2.get(ks, 1e30))*1e3:9.1f} us -> {conf.get(ks, 0)*1e3:9.1f} us  {keep[ks]}")
out_dir = os.path.join(ROOT, "learning-based-rgb-d-image-compression_amd", "csrc") if WRITE else os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
for blocked, fname in ((True, "tile_table_blk.h" if LATENCY else "tile_table_blk_loaded.h"), (False, "tile_table.h" if LATENCY else "tile_table_loaded.h")):
    src = os.path.join(ROOT, "learning-based-rgb-d-image-compression_amd", "csrc", fname)
    table, head = {}, []
    for ln in open(src):
        t = ln.strip()
        if t.startswith("{"):
            v = [int(x) for x in t.split("}")[0].strip("{},").replace(" ", "").split(",")]
            table[tuple(v[:9])] = tuple(v[9:])
        elif t.startswith("//"):
            head.append(ln.rstrip("\n"))
    for ks, c in keep.items():
        key = tuple(int(v) for v in ks.split(","))
        if (key[7] >= 100) == blocked:
            table[key] = c
    note = f"// + in-situ winners of tools/tune_insitu.py {B} {H} {W} {MODEL if MODEL != 'ELIC_united' else ''} (timed inside the codec call, one engine instance)".replace("  (", " (")
    if note not in head:
        head.append(note)
    with open(os.path.join(out_dir, ("" if WRITE else "insitu_") + fname), "w") as f:
        f.write("\n".join(head) + "\n")
        for key, c in sorted(table.items()):
            f.write("{" + ", ".join(str(v) for v in key) + ",   " + ", ".join(str(v) for v in c) + "},\n")
print("wrote tables to", out_dir)
