"""Float tensors of the reference's encoder (y, z, hyper parameters) for the golden inputs, as the CPU oracle computes them in
the survey container -- where tests/test_oracle_model.py pins the oracle to the unmodified reference bit for bit (same torch
CPU kernels).  The GPU's reference-arithmetic path is compared with these BITWISE in tests/test_gpu_refbits.py.

Small cases are stored whole; the 480x640 cases as per-channel hashes (a mismatch names its channels) + a 1/16 subsample.

container only:  python tests/golden/make_floats.py"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import elic_oracle as eo  # noqa: E402
from rgbd_amd import synth  # noqa: E402


def chan_hash(t):
    a = np.ascontiguousarray(t.numpy())
    # (+0.0 and -0.0 compare equal as floats: hash the bits of a + 0.0, which folds the sign of zero)
    a = a + np.float32(0.0)
    return np.stack([np.frombuffer(hashlib.sha1(a[:, c].tobytes()).digest()[:8], dtype=np.uint8) for c in range(a.shape[1])])


def main():
    torch.set_num_threads(8)
    cases = [("d_256x256", 0, "stress", True), ("g_256x256_s1", 1, "stress", True), ("h_256x256_s2", 2, "stress", True),
             ("f_480x640_stress", 0, "stress", False), ("e_480x640_tl", 0, "trained_like", False)]
    for name, seed, recipe, whole in cases:
        g = dict(np.load(os.path.join(HERE, f"model_{name}.npz")))
        sd = synth.synthetic_state_dict(seed) if recipe == "stress" else synth.synthetic_state_dict(seed, recipe=recipe)
        c = eo.OracleCodec(sd)
        c.update()
        r, d = synth.synthetic_batch(int(g["B"]), int(g["H"]), int(g["W"]), config_id=int(g["config_id"]))
        rp, dp = eo.pad_replicate0(torch.from_numpy(r)), eo.pad_replicate0(torch.from_numpy(d))
        c.trace = {}
        out = c.compress(rp, dp)
        assert out["r_strings"][0][0] == g["r_y"].tobytes() and out["d_strings"][0][0] == g["d_y"].tobytes(), name
        tr = c.trace
        keep = {}
        for k in ("y_r", "y_d", "z_r", "z_d", "hyper_r", "hyper_d"):
            t = tr[k]
            if whole:
                keep[k] = t.numpy()
            else:
                keep[k + "_hash"] = chan_hash(t)
                keep[k + "_sub"] = t.numpy()[:, :, ::4, ::4].copy()
        np.savez_compressed(os.path.join(HERE, f"floats_{name}.npz"), **keep)
        print(name, {k: v.shape for k, v in keep.items()})


if __name__ == "__main__":
    main()
