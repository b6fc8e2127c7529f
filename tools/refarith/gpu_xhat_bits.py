"""GPU box: the decoder's reconstruction against the reference's (tests/golden/model_*.npz: x_hat subsampled 8 x 8) -- how many
elements differ and by how much.  g_s's stride-2 transposed convs keep their single-chain sub-pixel form (no decision behind
them), so equality is not expected here: this prints what the difference is."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rgbd_amd
from rgbd_amd import ELIC_united, synth
from rgbd_amd.datautils import pad0
G = os.path.join(ROOT, "tests", "golden")
for name, seed, recipe in [("d_256x256", 0, None), ("f_480x640_stress", 0, None), ("e_480x640_tl", 0, "trained_like"), ("j_192x256_s3", 3, None),
                           ("k_200x300_tl_s4", 4, "trained_like")]:
    g = dict(np.load(os.path.join(G, f"model_{name}.npz")))
    sd = synth.synthetic_state_dict(seed) if recipe is None else synth.synthetic_state_dict(seed, recipe=recipe)
    net = ELIC_united(config=rgbd_amd.model_config(), channel=4).eval(); net.load_state_dict(sd); net.update(force=True); net = net.to("cuda")
    H, W = int(g["H"]), int(g["W"])
    r, d = synth.synthetic_batch(1, H, W, config_id=int(g["config_id"]))
    rp, dp = pad0(torch.from_numpy(r), mode="replicate"), pad0(torch.from_numpy(d), mode="replicate")
    out = net.compress(rp.cuda(), dp.cuda())
    rec = net.decompress(out["r_strings"], out["d_strings"], out["shape"])
    for m, key in (("r", "xhat_r_sub"), ("d", "xhat_d_sub")):
        x = rec["x_hat"][m].cpu()[:, :, :H, :W][:, :, ::8, ::8].numpy()
        ref = g[key]
        bad = x != ref
        print(name, m, "differ", int(bad.sum()), "of", x.size, "max abs", float(np.abs(x - ref).max()), flush=True)
    del net
