"""The encoder's float tensors -- y, z and the hyper parameters, i.e. everything a coding decision is taken from before the
Bi-CEE loop -- BITWISE against the reference's (tests/golden/floats_*.npz: the CPU oracle in the survey container, where
tests/test_oracle_model.py pins it to the unmodified reference bit for bit; generator tests/golden/make_floats.py).

This is the float half of "streams identical" (DESIGN.md 4a): the GPU runs every such operation in the accumulation order
and with the roundings of the CPU kernel the reference runs it on (analysis.py:116-174, 231-242; synthesis.py:305-323), so
the tensors are expected to be EQUAL, element for element, not close.  Small cases are stored whole; the 480x640 cases as
per-channel hashes (a mismatch names its channels) plus a 1/16 subsample (which bounds how far off a mismatch is)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from gpu_utils import require_gpu

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TENSORS = ("y_r", "y_d", "z_r", "z_d", "hyper_r", "hyper_d")


def _chan_hash(a):
    a = np.ascontiguousarray(a) + np.float32(0.0)  # (folds the sign of zero, as make_floats.py does)
    return np.stack([np.frombuffer(hashlib.sha1(a[:, c].tobytes()).digest()[:8], dtype=np.uint8) for c in range(a.shape[1])])


@pytest.mark.parametrize("name,seed,recipe", [("d_256x256", 0, "stress"), ("g_256x256_s1", 1, "stress"), ("h_256x256_s2", 2, "stress"),
                                              ("f_480x640_stress", 0, "stress"), ("e_480x640_tl", 0, "trained_like")])
def test_encoder_floats_equal_the_references_bit_for_bit(name, seed, recipe):
    require_gpu()
    import rgbd_amd
    from rgbd_amd import ELIC_united, synth
    from rgbd_amd._lib import lib
    from rgbd_amd.datautils import pad0

    g = dict(np.load(os.path.join(GOLDEN, f"model_{name}.npz")))
    fl = dict(np.load(os.path.join(GOLDEN, f"floats_{name}.npz")))
    sd = synth.synthetic_state_dict(seed) if recipe == "stress" else synth.synthetic_state_dict(seed, recipe=recipe)
    net = ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
    net.load_state_dict(sd)
    net.update(force=True)
    net = net.to("cuda")
    r, d = synth.synthetic_batch(int(g["B"]), int(g["H"]), int(g["W"]), config_id=int(g["config_id"]))
    rp, dp = pad0(torch.from_numpy(r), mode="replicate"), pad0(torch.from_numpy(d), mode="replicate")
    out = net.compress(rp.cuda(), dp.cuda())
    assert lib().rgbd_elic_ref_table_misses(net._h) == 0, "a layer shape of this case has no entry in refarith_tables.json"
    for k in TENSORS:
        t = net.debug_tensor(k)
        if k in fl:
            bad = (t + np.float32(0.0)) != (fl[k] + np.float32(0.0))
            assert not bad.any(), (name, k, int(bad.sum()), t.size, float(np.abs(t - fl[k]).max() / np.abs(fl[k]).max()))
        else:
            bad = (_chan_hash(t) != fl[k + "_hash"]).any(axis=1)
            sub = t[:, :, ::4, ::4]
            assert not bad.any(), (name, k, "channels that differ:", np.nonzero(bad)[0][:16].tolist(),
                                   float(np.abs(sub - fl[k + "_sub"]).max() / np.abs(fl[k + "_sub"]).max()))
    # ... and then the streams are the reference's (also asserted, clause by clause, in test_gpu_parity_pinned.py)
    assert out["r_strings"][0][0] == g["r_y"].tobytes() and out["d_strings"][0][0] == g["d_y"].tobytes()
    assert out["r_strings"][1][0] == g["r_z0"].tobytes() and out["d_strings"][1][0] == g["d_z0"].tobytes()
