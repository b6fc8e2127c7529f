"""GPU box: WHERE the encoder's float tensors differ from the reference's (tests/golden/floats_*.npz), element by element for the
small cases, channel by channel for the 480x640 ones.  Development aid behind tests/test_gpu_refbits.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rgbd_amd
from rgbd_amd import ELIC_united, synth
from rgbd_amd.datautils import pad0
from test_gpu_refbits import _chan_hash, TENSORS
G = os.path.join(ROOT, "tests", "golden")
for name, seed, recipe in [("d_256x256", 0, "stress"), ("g_256x256_s1", 1, "stress"), ("h_256x256_s2", 2, "stress"),
                           ("f_480x640_stress", 0, "stress"), ("e_480x640_tl", 0, "trained_like")]:
    g = dict(np.load(os.path.join(G, f"model_{name}.npz"))); fl = dict(np.load(os.path.join(G, f"floats_{name}.npz")))
    sd = synth.synthetic_state_dict(seed) if recipe == "stress" else synth.synthetic_state_dict(seed, recipe=recipe)
    net = ELIC_united(config=rgbd_amd.model_config(), channel=4).eval(); net.load_state_dict(sd); net.update(force=True); net = net.to("cuda")
    r, d = synth.synthetic_batch(int(g["B"]), int(g["H"]), int(g["W"]), config_id=int(g["config_id"]))
    rp, dp = pad0(torch.from_numpy(r), mode="replicate"), pad0(torch.from_numpy(d), mode="replicate")
    net.compress(rp.cuda(), dp.cuda())
    for k in TENSORS:
        t = net.debug_tensor(k)
        if k in fl:
            bad = np.argwhere((t + np.float32(0)) != (fl[k] + np.float32(0)))
            print(name, k, len(bad), "of", t.size, [(tuple(int(v) for v in b), float(t[tuple(b)]), float(fl[k][tuple(b)])) for b in bad[:6]], flush=True)
        else:
            badc = np.nonzero((_chan_hash(t) != fl[k + "_hash"]).any(axis=1))[0]
            print(name, k, "channels differing:", len(badc), badc[:12].tolist(), flush=True)
    del net
