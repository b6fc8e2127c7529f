"""GPU box: how many bits of the encoder's float tensors (y, z, hyper) equal the reference's (tests/golden/floats_*.npz),
and do the streams equal the golden streams?  Development aid behind tests/test_gpu_refbits.py."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rgbd_amd
from rgbd_amd import ELIC_united, synth
G = os.path.join(ROOT, "tests", "golden")

def chan_hash(a):
    a = np.ascontiguousarray(a) + np.float32(0.0)
    return np.stack([np.frombuffer(hashlib.sha1(a[:, c].tobytes()).digest()[:8], dtype=np.uint8) for c in range(a.shape[1])])

def run(name, seed, recipe):
    g = dict(np.load(os.path.join(G, f"model_{name}.npz")))
    fl = dict(np.load(os.path.join(G, f"floats_{name}.npz")))
    sd = synth.synthetic_state_dict(seed) if recipe == "stress" else synth.synthetic_state_dict(seed, recipe=recipe)
    net = ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
    net.load_state_dict(sd); net.update(force=True); net = net.to("cuda:0")
    r, d = synth.synthetic_batch(int(g["B"]), int(g["H"]), int(g["W"]), config_id=int(g["config_id"]))
    from rgbd_amd.datautils import pad0
    rp, dp = pad0(torch.from_numpy(r), mode="replicate"), pad0(torch.from_numpy(d), mode="replicate")
    out = net.compress(rp.cuda(), dp.cuda())
    line = [name]
    for k in ("y_r", "y_d", "z_r", "z_d", "hyper_r", "hyper_d"):
        t = net.debug_tensor(k)
        if k in fl:
            ref = fl[k]; bad = t != ref
            line.append(f"{k}: {int(bad.sum())}/{t.size} differ, max rel {np.abs(t-ref).max()/np.abs(ref).max():.2e}")
        else:
            bad = (chan_hash(t) != fl[k + "_hash"]).any(axis=1)
            sub = t[:, :, ::4, ::4]
            line.append(f"{k}: {int(bad.sum())}/{bad.size} channels differ, sub max rel {np.abs(sub-fl[k+'_sub']).max()/np.abs(fl[k+'_sub']).max():.2e}")
    same = [out["r_strings"][0][0] == g["r_y"].tobytes(), out["d_strings"][0][0] == g["d_y"].tobytes(),
            out["r_strings"][1][0] == g["r_z0"].tobytes(), out["d_strings"][1][0] == g["d_z0"].tobytes()]
    line.append(f"streams equal (r_y, d_y, r_z, d_z): {same}  len r_y {len(out['r_strings'][0][0])} vs {g['r_y'].size}")
    print("\n   ".join(line), flush=True)
    from rgbd_amd._lib import lib
    print("   table misses:", lib().rgbd_elic_ref_table_misses(net._h))

if __name__ == "__main__":
    cases = [("d_256x256", 0, "stress"), ("g_256x256_s1", 1, "stress"), ("h_256x256_s2", 2, "stress"),
             ("f_480x640_stress", 0, "stress"), ("e_480x640_tl", 0, "trained_like")]
    sel = sys.argv[1:] 
    for c in cases:
        if not sel or c[0] in sel: run(*c)
