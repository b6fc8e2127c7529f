#!/usr/bin/env python3
"""How close to a decision boundary was the reference when its stream and the GPU's part?  

Runs the UNMODIFIED reference in this container (tests/golden/_reference_loader.py) with observers on the three
decisions that turn floats into integers --

    GaussianConditional.quantize(y, "symbols", means)   round(y - mean)      entropy_models.py:131-137
    GaussianConditional.build_indexes(scales)            scale -> table row   entropy_models.py:561-568
    EntropyBottleneck.quantize(z, "symbols", medians)    round(z - median)    entropy_models.py:437-440

-- and stores, per golden case, the reference's symbols / indexes in stream order plus every symbol whose rounded value
or scale lies within a small window of a boundary ("near-boundary" list: position, y - mean, scale), and all z - median
values (they are few).  tests/test_gpu_parity_pinned.py finds the first symbol where the GPU's (symbol, index) sequence
leaves the reference's, requires it to be on that list, and records both margins: the reference's distance to the boundary
and the GPU's float difference at that element.  "The flip is inherent to fp32 summation order" then is a number.

Existing goldens are not rewritten: the run must reproduce their streams byte for byte (else this container's CPU path
differs from the one that made them and the script stops).  New end-to-end cases (stress recipe at the bench's 480x640
shape; weight seeds 1 and 2 at 256x256) get their model_*.npz here as well.

    make -C oracle ref && python tests/golden/make_margins.py [case ...]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _reference_loader as rl  # noqa: E402
import make_golden as mg  # noqa: E402

ROUND_WINDOW = 2e-4   # |distance to .5| <= ROUND_WINDOW * max(1, |y - mean|)
SCALE_WINDOW = 2e-4   # |scale / table_entry - 1| <= SCALE_WINDOW


class Observer:
    """Wraps the three integer decisions of one modality's entropy models while `on`."""

    def __init__(self, gc, eb):
        self.on = False
        self.x, self.s, self.sym, self.idx, self.zx, self.zsym = [], [], [], [], [], []
        self.table = gc.scale_table.detach().numpy().astype(np.float32)
        q0, b0 = gc.quantize, gc.build_indexes

        def quantize(inputs, mode, means=None):
            out = q0(inputs, mode, means)
            if self.on and mode == "symbols":
                self.x.append((inputs - means).reshape(-1).numpy().astype(np.float32).copy())
                self.sym.append(out.reshape(-1).numpy().astype(np.int32).copy())
            return out

        def build_indexes(scales):
            out = b0(scales)
            if self.on:
                self.s.append(scales.reshape(-1).numpy().astype(np.float32).copy())
                self.idx.append(out.reshape(-1).numpy().astype(np.int32).copy())
            return out

        gc.quantize, gc.build_indexes = quantize, build_indexes
        if eb is not None:
            zq0 = eb.quantize

            def zquantize(inputs, mode, means=None):
                out = zq0(inputs, mode, means)
                if self.on and mode == "symbols":
                    self.zx.append((inputs - means).reshape(-1).numpy().astype(np.float32).copy())
                    self.zsym.append(out.reshape(-1).numpy().astype(np.int32).copy())
                return out

            eb.quantize = zquantize

    def pack(self, tag):
        x, s = np.concatenate(self.x), np.concatenate(self.s)
        sym, idx = np.concatenate(self.sym), np.concatenate(self.idx)
        assert x.shape == s.shape == sym.shape == idx.shape
        m_round = 0.5 - np.abs(x - np.rint(x))  # distance of y - mean to the nearest rounding boundary
        # (a scale below LowerBound(0.11), entropy_models.py:562, is clamped onto table[0]: only its distance to 0.11 matters)
        rel = np.abs(s[:, None] / self.table[None, :-1] - 1.0).min(axis=1)
        near = (m_round <= ROUND_WINDOW * np.maximum(1.0, np.abs(x))) | (rel <= SCALE_WINDOW)
        pos = np.nonzero(near)[0].astype(np.int32)
        out = {f"ref_sym_{tag}": sym.astype(np.int16 if np.abs(sym).max() < 32000 else np.int32),
               f"ref_idx_{tag}": idx.astype(np.uint8), f"nb_pos_{tag}": pos, f"nb_x_{tag}": x[pos], f"nb_s_{tag}": s[pos],
               f"parts_{tag}": np.array([len(a) for a in self.x], np.int64)}
        if self.zx:
            out[f"ref_zx_{tag}"] = np.concatenate(self.zx)
            out[f"ref_zsym_{tag}"] = np.concatenate(self.zsym).astype(np.int16)
        return out


def observe(net, prefixes):
    obs = {}
    for tag, pre in prefixes.items():
        gc = getattr(net, pre + "gaussian_conditional")
        eb = getattr(net, pre + "entropy_bottleneck", None)
        obs[tag] = Observer(gc, eb)
    return obs


def run(obs, fn):
    for o in obs.values():
        o.on = True
    try:
        with torch.no_grad():
            return fn()
    finally:
        for o in obs.values():
            o.on = False


def save(name, obs, extra=None):
    g = {"round_window": ROUND_WINDOW, "scale_window": SCALE_WINDOW}
    for tag, o in obs.items():
        g.update(o.pack(tag))
    g.update(extra or {})
    path = os.path.join(HERE, f"margins_{name}.npz")
    np.savez_compressed(path, **g)
    print("margins", name, {k: v.shape for k, v in g.items() if hasattr(v, "shape") and v.ndim}, os.path.getsize(path))


def same_streams(path, pairs):
    g = np.load(path)
    for key, got in pairs:
        if g[key].tobytes() != got:
            raise SystemExit(f"{os.path.basename(path)}: {key} differs from what this container's reference run produces "
                             "(another CPU / oneDNN path than the one that made the golden): not rewriting anything")


def united_case(ELIC, model_config, synth, name, B, H, W, cid, seed=0, recipe=None, new=False, smooth=False):
    from dataset.utils import pad

    net = ELIC(config=model_config(), channel=4).eval()
    sd = synth.synthetic_state_dict(seed) if recipe is None else synth.synthetic_state_dict(seed, recipe=recipe)
    net.load_state_dict(sd)
    assert net.update(force=True)
    if new:
        mg.model_case(net, synth, name, B, H, W, cid, False, smooth=smooth)
    obs = observe(net, {"r": "rgb_", "d": "depth_"})
    r, d = synth.synthetic_batch(B, H, W, config_id=cid, smooth=smooth)
    rp, dp = pad(torch.from_numpy(r), "replicate0"), pad(torch.from_numpy(d), "replicate0")
    out = run(obs, lambda: net.compress(rp, dp))
    same_streams(os.path.join(HERE, f"model_{name}.npz"), [("r_y", out["r_strings"][0][0]), ("d_y", out["d_strings"][0][0]),
                                                          ("r_z0", out["r_strings"][1][0]), ("d_z0", out["d_strings"][1][0])])
    save(name, obs, {"weights_seed": seed})


def bicee_case(ELIC, model_config, synth, name, B, h, w, seed):
    net = ELIC(config=model_config(), channel=4).eval()
    net.load_state_dict(synth.synthetic_state_dict(0))
    assert net.update(force=True)
    obs = observe(net, {"r": "rgb_", "d": "depth_"})
    yr, hr, yd, hd = [torch.from_numpy(a) for a in synth.synthetic_latents(B, h, w, 320, seed)]
    sr, sdp = run(obs, lambda: net.compress_united(yr, hr, yd, hd))
    same_streams(os.path.join(HERE, f"bicee_{name}.npz"), [("r_y", sr[0]), ("d_y", sdp[0])])
    save("bicee_" + name, obs)


def single_case(ext, model_config, synth, name, H, W, cid):
    net = ext["ELIC"](config=model_config(), channel=3).eval()
    net.load_state_dict(synth.synthetic_state_dict(0, model="ELIC"))
    assert net.update(force=True)
    obs = observe(net, {"r": ""})
    r, _ = synth.synthetic_batch(1, H, W, config_id=cid)
    out = run(obs, lambda: net.compress(torch.from_numpy(r)))
    same_streams(os.path.join(HERE, f"elic_{name}.npz"), [("y_stream", out["strings"][0][0]), ("z0", out["strings"][1][0])])
    save("elic_" + name, obs)


def stf_case(model_config, synth, name, H, W, cid):
    from models.stf_united import SymmetricalTransFormerUnited as STF

    net = STF(config=model_config(), channel=4).eval()
    net.load_state_dict(synth.synthetic_state_dict(0, model="STF_united"))
    assert net.update(force=True)
    obs = observe(net, {"r": "rgb_", "d": "depth_"})
    r, d = synth.synthetic_batch(1, H, W, config_id=cid)
    out = run(obs, lambda: net.compress(torch.from_numpy(r), torch.from_numpy(d)))
    same_streams(os.path.join(HERE, f"stf_{name}.npz"), [("r_y", out["r_strings"][0][0]), ("d_y", out["d_strings"][0][0]),
                                                        ("r_z0", out["r_strings"][1][0]), ("d_z0", out["d_strings"][1][0])])
    save("stf_" + name, obs)


def main():
    ELIC, model_config, ext = rl.load_reference()
    import rgbd_amd  # noqa: F401
    from rgbd_amd import synth

    want = set(sys.argv[1:])

    def on(n):
        return not want or n in want

    if on("d_256x256"):
        united_case(ELIC, model_config, synth, "d_256x256", 1, 256, 256, 2)
    if on("e_480x640_tl"):
        united_case(ELIC, model_config, synth, "e_480x640_tl", 1, 480, 640, 3, recipe="trained_like")
    # new end-to-end goldens: the bench's operating point (stress recipe at 480x640), and the flip census seeds at 256x256
    if on("f_480x640_stress"):
        united_case(ELIC, model_config, synth, "f_480x640_stress", 1, 480, 640, 3, new=True)
    if on("g_256x256_s1"):
        united_case(ELIC, model_config, synth, "g_256x256_s1", 1, 256, 256, 2, seed=1, new=True)
    if on("h_256x256_s2"):
        united_case(ELIC, model_config, synth, "h_256x256_s2", 1, 256, 256, 2, seed=2, new=True)
    # round 5: a HELD-OUT case -- an image size (192x256) and a weight seed (3) no table entry, kernel or test had seen when the
    # reference-arithmetic path was written; its layer shapes were then measured with `tools/refarith/discover.py --add
    # united:192:256:1` and nothing else changed.  (Named explicitly only: not part of the default run.)
    if "j_192x256_s3" in want:
        united_case(ELIC, model_config, synth, "j_192x256_s3", 1, 192, 256, 12, seed=3, new=True)
    if "l_b2_192x256_s5" in want:  # third held-out case: the reference's batched calling convention (one stream per batch of two)
        united_case(ELIC, model_config, synth, "l_b2_192x256_s5", 2, 192, 256, 14, seed=5, new=True)
    if "m_256x320_smooth_s7" in want:  # fourth held-out case: spatially correlated images (every other golden codes uniform noise)
        united_case(ELIC, model_config, synth, "m_256x320_smooth_s7", 1, 256, 320, 15, seed=7, new=True, smooth=True)
    if "p_480x640_s10" in want:  # held-out at the bench's own image shape: another weight seed (10) and other images
        united_case(ELIC, model_config, synth, "p_480x640_s10", 1, 480, 640, 18, seed=10, new=True)
    if "k_200x300_tl_s4" in want:  # a second held-out case: a size that needs padding (-> 256 x 320), trained-like weights, seed 4
        united_case(ELIC, model_config, synth, "k_200x300_tl_s4", 1, 200, 300, 13, seed=4, recipe="trained_like", new=True)
    if on("i_128x192_hr"):  # the high_rate weights (wide CDF rows); golden from make_golden.py --only-hr
        united_case(ELIC, model_config, synth, "i_128x192_hr", 1, 128, 192, 41, recipe="high_rate")
    if on("bicee_c4_b2_8x12"):
        bicee_case(ELIC, model_config, synth, "c4_b2_8x12", 2, 8, 12, 5)
    if on("elic_c1_256x256"):
        single_case(ext, model_config, synth, "c1_256x256", 256, 256, 1)
    if on("stf_c5_256x256"):
        stf_case(model_config, synth, "c5_256x256", 256, 256, 5)


if __name__ == "__main__":
    main()
