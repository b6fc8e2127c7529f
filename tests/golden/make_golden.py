#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE UNMODIFIED REFERENCE in this container.

    make -C oracle ref && python tests/golden/make_golden.py

Outputs (data only -- inputs are regenerated from rgbd_amd.synth, never stored):
    tests/golden/coder_kat.npz      pure-coder known answers from the reference's C++ (KAT tiny / B2 / tables)
    tests/golden/model_*.npz        ELIC_united streams, latents and reconstructions from the reference's
                                    compress()/decompress() on synthetic weights + inputs (model_e_480x640_tl: the bench's
                                    image shape with the trained_like weights, `--only-e`; model_i_128x192_hr: the high_rate
                                    weights, `--only-hr`)
    tests/golden/bicee_*.npz        Bi-CEE stage alone (BASELINE config 4): compress_united / decompress_united outputs
    tests/golden/elic_*.npz         single-modal ELIC (BASELINE config 1): streams, latents, reconstruction
    tests/golden/stf_*.npz          STF_united (Swin transforms; BASELINE config 5 at reduced size)
    tests/golden/r2d_*.npz          ELIC_united_R2D (one-directional variant)
    tests/golden/harness.json       pad / container / bpp / PSNR tuples (TesterUnited arithmetic)

The reference runs on PyTorch CPU kernels; float tensors are therefore specific to this container's
CPU/oneDNN build.  Integer fixtures (tables, streams given symbols) are machine independent.
"""
import hashlib
import io
import json
import os
import struct
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _reference_loader as rl  # noqa: E402


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()[:16]


def coder_kats(net, ext):
    ans, cxx = ext["ans"], ext["_CXX"]
    gc = net.rgb_gaussian_conditional
    cdf = gc._quantized_cdf.numpy().astype(np.int32)
    sizes = gc._cdf_length.numpy().astype(np.int32)
    offs = gc._offset.numpy().astype(np.int32)
    cdf_l, sizes_l, offs_l = cdf.tolist(), sizes.tolist(), offs.tolist()
    table = gc.scale_table.numpy().astype(np.float32)
    out = {"gc_cdf": cdf, "gc_sizes": sizes, "gc_offsets": offs, "scale_table": table}

    # tiny hand-checkable KAT
    sym = [0, 1, -1, 0, 2, -3, 0, 0, 5, -7, 40, -40]
    idx = [0, 0, 0, 1, 5, 5, 10, 10, 10, 10, 0, 3]
    s = ans.RansEncoder().encode_with_indexes(sym, idx, cdf_l, sizes_l, offs_l)
    out["tiny_sym"] = np.array(sym, np.int32)
    out["tiny_idx"] = np.array(idx, np.int32)
    out["tiny_stream"] = np.frombuffer(s, np.uint8)
    enc = ans.BufferedRansEncoder()
    enc.encode_with_indexes(sym[:5], idx[:5], cdf_l, sizes_l, offs_l)
    enc.encode_with_indexes(sym[5:], idx[5:], cdf_l, sizes_l, offs_l)
    assert enc.flush() == s
    dec = ans.RansDecoder()
    dec.set_stream(s)
    got = dec.decode_stream(idx[:4], cdf_l, sizes_l, offs_l) + dec.decode_stream(idx[4:9], cdf_l, sizes_l, offs_l) \
        + dec.decode_stream(idx[9:], cdf_l, sizes_l, offs_l)
    assert got == sym

    # KAT-B2: 49,152 symbols with escapes (numpy legacy RandomState is frozen => reproducible anywhere)
    rng = np.random.RandomState(1234)
    n = 49152
    bidx = rng.randint(0, 64, n)
    bsym = np.rint(rng.standard_normal(n) * table[bidx]).astype(np.int64)
    bsym[::97] *= 8
    bsym[5::193] = -bsym[5::193] - 3
    bsym = bsym.astype(np.int32)
    bs = ans.RansEncoder().encode_with_indexes(bsym.tolist(), bidx.astype(np.int32).tolist(), cdf_l, sizes_l, offs_l)
    assert ans.RansDecoder().decode_with_indexes(bs, bidx.tolist(), cdf_l, sizes_l, offs_l) == bsym.tolist()
    out["b2_stream"] = np.frombuffer(bs, np.uint8)
    out["b2_sym_sha"] = np.frombuffer(sha(bsym.tobytes()).encode(), np.uint8)
    out["b2_idx_sha"] = np.frombuffer(sha(bidx.astype(np.int32).tobytes()).encode(), np.uint8)

    # (empty and 1-symbol inputs under-allocate in the reference's flush(), rans_interface.cpp:171 -- undefined
    #  behaviour there, so no known answer is taken from it; the build defines them: see tests/test_oracle_coder.py)
    out["two_stream"] = np.frombuffer(ans.RansEncoder().encode_with_indexes([3, -2], [20, 7], cdf_l, sizes_l, offs_l),
                                      np.uint8)
    # all-escape stream on the narrowest row
    esym = (np.arange(-300, 300, 7)).astype(np.int32)
    es = ans.RansEncoder().encode_with_indexes(esym.tolist(), [0] * len(esym), cdf_l, sizes_l, offs_l)
    out["esc_sym"] = esym
    out["esc_stream"] = np.frombuffer(es, np.uint8)

    # pmf_to_quantized_cdf known answers, incl. one needing the steal loop
    pm = [[0.1, 0.2, 0.7], [1e-9, 1 - 2e-9, 1e-9], [0.25] * 4, [1e-7] * 5 + [1.0 - 5e-7] + [1e-7] * 5]
    for k, p in enumerate(pm):
        out[f"pmf{k}"] = np.array(p, np.float32)
        out[f"pmf{k}_cdf"] = np.array(cxx.pmf_to_quantized_cdf([float(np.float32(v)) for v in p], 16), np.uint32)
    return out


def eb_tables(net):
    out = {}
    for m in ("rgb", "depth"):
        eb = getattr(net, f"{m}_entropy_bottleneck")
        out[f"{m}_eb_cdf"] = eb._quantized_cdf.numpy().astype(np.int32)
        out[f"{m}_eb_sizes"] = eb._cdf_length.numpy().astype(np.int32)
        out[f"{m}_eb_offsets"] = eb._offset.numpy().astype(np.int32)
    return out


def container(H, W, shape, strings) -> bytes:
    # what TesterUnited writes through the reference's IOutils (tester_united.py:153-165)
    from utils.IOutils import write_body, write_uints

    f = io.BytesIO()
    write_uints(f, (H, W))
    write_body(f, shape, strings)
    return f.getvalue()


def model_case(net, synth, name, B, H, W, config_id, full: bool, smooth: bool = False):
    from dataset.utils import crop0, pad  # reference harness pieces that import cleanly

    r, d = synth.synthetic_batch(B, H, W, config_id=config_id, smooth=smooth)
    r, d = torch.from_numpy(r), torch.from_numpy(d)
    rp, dp = pad(r, "replicate0"), pad(d, "replicate0")
    with torch.no_grad():
        out = net.compress(rp, dp)
        dec = net.decompress(out["r_strings"], out["d_strings"], out["shape"])
    xr, xd = crop0(dec["x_hat"]["r"], (H, W)), crop0(dec["x_hat"]["d"], (H, W))
    g = {"B": B, "H": H, "W": W, "config_id": config_id, "shape": np.array(tuple(out["shape"]), np.int32),
         "padded": np.array(rp.shape[-2:], np.int32)}
    if smooth:
        g["smooth"] = 1  # spatially correlated inputs (synth.synthetic_pair(smooth=True)) instead of uniform noise
    for m, key in (("r", "r_strings"), ("d", "d_strings")):
        g[f"{m}_y"] = np.frombuffer(out[key][0][0], np.uint8)
        for i, s in enumerate(out[key][1]):
            g[f"{m}_z{i}"] = np.frombuffer(s, np.uint8)
    if B == 1:
        cr = container(H, W, out["shape"], out["r_strings"])
        cd = container(H, W, out["shape"], out["d_strings"])
        g["r_container_sha"] = np.frombuffer(sha(cr).encode(), np.uint8)
        g["d_container_sha"] = np.frombuffer(sha(cd).encode(), np.uint8)
        g["bpp"] = np.array([len(cr) * 8.0 / (H * W), len(cd) * 8.0 / (H * W)], np.float64)
    mse_r = torch.mean((xr.clamp(0, 1) - r.clamp(0, 1)) ** 2).item()
    mse_d = torch.mean((xd.clamp(0, 1) - d.clamp(0, 1)) ** 2).item()
    g["psnr"] = np.array([-10 * np.log10(mse_r), -10 * np.log10(mse_d)], np.float64)
    g["xhat_r_mean"] = np.array([xr.double().mean().item(), xd.double().mean().item()], np.float64)
    if full:
        with torch.no_grad():
            y_r, y_d = net.g_a(rp, dp)
            z_r, z_d = net.h_a(y_r, y_d)
            zh_r = net.rgb_entropy_bottleneck.decompress(out["r_strings"][1], out["shape"])
            zh_d = net.depth_entropy_bottleneck.decompress(out["d_strings"][1], out["shape"])
            hp_r, hp_d = net.h_s(zh_r, zh_d)
            fw = net(rp, dp)
        assert torch.equal(fw["x_hat"]["r"].clamp(0, 1), dec["x_hat"]["r"])
        g.update({"fw_xhat_r": fw["x_hat"]["r"].numpy(), "fw_xhat_d": fw["x_hat"]["d"].numpy(),
                  "lik_y_r": fw["r_likelihoods"]["y"].numpy(), "lik_y_d": fw["d_likelihoods"]["y"].numpy(),
                  "lik_z_r": fw["r_likelihoods"]["z"].numpy(), "lik_z_d": fw["d_likelihoods"]["z"].numpy()})
        g.update({"y_r": y_r.numpy(), "y_d": y_d.numpy(), "z_r": z_r.numpy(), "z_d": z_d.numpy(),
                  "zhat_r": zh_r.numpy(), "zhat_d": zh_d.numpy(), "hyper_r": hp_r.numpy(), "hyper_d": hp_d.numpy(),
                  "xhat_r": xr.numpy(), "xhat_d": xd.numpy()})
    else:
        g["xhat_r_sub"] = xr[:, :, ::8, ::8].numpy()
        g["xhat_d_sub"] = xd[:, :, ::8, ::8].numpy()
    np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **g)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in g.items() if k[0] in "rd" and "_" in k[:3]},
          g["psnr"], g.get("bpp"))
    return g


def bicee_case(net, synth, name, B, h, w, seed):
    """BASELINE config 4: the reference's compress_united / decompress_united on given latents + hyper parameters."""
    yr, hr, yd, hd = [torch.from_numpy(a) for a in synth.synthetic_latents(B, h, w, 320, seed)]
    with torch.no_grad():
        sr, sdp = net.compress_united(yr, hr, yd, hd)
        yhat_r, yhat_d = net.decompress_united(sr[0], hr, sdp[0], hd)
    g = {"B": B, "h": h, "w": w, "seed": seed, "r_y": np.frombuffer(sr[0], np.uint8), "d_y": np.frombuffer(sdp[0], np.uint8),
         "yhat_r": yhat_r.numpy(), "yhat_d": yhat_d.numpy()}
    np.savez_compressed(os.path.join(HERE, f"bicee_{name}.npz"), **g)
    print("bicee", name, len(sr[0]), len(sdp[0]), float(yhat_r.abs().mean()))
    return g


def elic_single_case(ext, model_config, synth, name, B, H, W, config_id, seed=0):
    """BASELINE config 1: the reference's single-modal ELIC (models/elic.py) compress()/decompress() on one RGB image."""
    net = ext["ELIC"](config=model_config(), channel=3).eval()
    net.load_state_dict(synth.synthetic_state_dict(seed, model="ELIC"))
    assert net.update(force=True)
    r, _ = synth.synthetic_batch(B, H, W, config_id=config_id)
    x = torch.from_numpy(r)
    with torch.no_grad():
        out = net.compress(x)
        dec = net.decompress(out["strings"], out["shape"])
        y = net.g_a(x)
        z = net.h_a(y)
        hyper = net.h_s(net.entropy_bottleneck.decompress(out["strings"][1], out["shape"]))
    g = {"B": B, "H": H, "W": W, "config_id": config_id, "shape": np.array(tuple(out["shape"]), np.int32),
         "y_stream": np.frombuffer(out["strings"][0][0], np.uint8), "y": y.numpy(), "z": z.numpy(), "hyper": hyper.numpy(),
         "xhat_sub": dec["x_hat"][:, :, ::4, ::4].numpy(),
         "psnr": np.array([-10 * np.log10(torch.mean((dec["x_hat"].clamp(0, 1) - x) ** 2).item())], np.float64)}
    for i, zs in enumerate(out["strings"][1]):
        g[f"z{i}"] = np.frombuffer(zs, np.uint8)
    g["eb_cdf"] = net.entropy_bottleneck._quantized_cdf.numpy().astype(np.int32)
    np.savez_compressed(os.path.join(HERE, f"elic_{name}.npz"), **g)
    print("elic", name, len(out["strings"][0][0]), len(out["strings"][1][0]), g["psnr"])
    return g


def stf_case(model_config, synth, name, B, H, W, config_id):
    """BASELINE config 5 (reduced size): the reference's STF_united (models/stf_united.py) compress()/decompress()."""
    from models.stf_united import SymmetricalTransFormerUnited as STF

    net = STF(config=model_config(), channel=4).eval()
    net.load_state_dict(synth.synthetic_state_dict(0, model="STF_united"))
    assert net.update(force=True)
    r, d = synth.synthetic_batch(B, H, W, config_id=config_id)
    r, d = torch.from_numpy(r), torch.from_numpy(d)
    with torch.no_grad():
        y_r, y_d = net.g_a(r, d)
        out = net.compress(r, d)
        dec = net.decompress(out["r_strings"], out["d_strings"], out["shape"])
        fw = net(r, d)  # eval-mode forward (inherited models/elic_united.py:234-263 over the R2D slice coder)
        # the Bi-CEE stage alone (inherited compress_united / decompress_united, elic_united.py:350-401,543-578)
        lat = [torch.from_numpy(a) for a in synth.synthetic_latents(1, 8, 12, 320, 6)]
        cu = net.compress_united(lat[0], lat[1], lat[2], lat[3])
        du = net.decompress_united(cu[0][0], lat[1], cu[1][0], lat[3])
    g = {"B": B, "H": H, "W": W, "config_id": config_id, "shape": np.array(tuple(out["shape"]), np.int32),
         "r_y": np.frombuffer(out["r_strings"][0][0], np.uint8), "d_y": np.frombuffer(out["d_strings"][0][0], np.uint8),
         "r_z0": np.frombuffer(out["r_strings"][1][0], np.uint8), "d_z0": np.frombuffer(out["d_strings"][1][0], np.uint8),
         "y_r": y_r.numpy(), "y_d": y_d.numpy(),
         "fw_xhat_r_sub": fw["x_hat"]["r"][:, :, ::4, ::4].numpy(), "fw_xhat_d_sub": fw["x_hat"]["d"][:, :, ::4, ::4].numpy(),
         "lik_y_r": fw["r_likelihoods"]["y"].numpy(), "lik_y_d": fw["d_likelihoods"]["y"].numpy(),
         "lik_z_r": fw["r_likelihoods"]["z"].numpy(), "lik_z_d": fw["d_likelihoods"]["z"].numpy(),
         "cu_seed": 6, "cu_r_y": np.frombuffer(cu[0][0], np.uint8), "cu_d_y": np.frombuffer(cu[1][0], np.uint8),
         "cu_yhat_r": du[0].numpy(), "cu_yhat_d": du[1].numpy(),
         "xhat_r_sub": dec["x_hat"]["r"][:, :, ::4, ::4].numpy(), "xhat_d_sub": dec["x_hat"]["d"][:, :, ::4, ::4].numpy(),
         "psnr": np.array([-10 * np.log10(torch.mean((dec["x_hat"]["r"] - r) ** 2).item()),
                           -10 * np.log10(torch.mean((dec["x_hat"]["d"] - d) ** 2).item())], np.float64)}
    np.savez_compressed(os.path.join(HERE, f"stf_{name}.npz"), **g)
    print("stf", name, len(out["r_strings"][0][0]), len(out["d_strings"][0][0]), g["psnr"])
    return g


def r2d_case(model_config, synth, name, B, H, W, config_id, seed=0):
    """the reference's ELIC_united_R2D (models/elic_united_R2D.py) compress()/decompress()."""
    from models.elic_united_R2D import ELIC_united_R2D

    net = ELIC_united_R2D(config=model_config(), channel=4).eval()
    net.load_state_dict(synth.synthetic_state_dict(seed, model="ELIC_united_R2D"))
    assert net.update(force=True)
    r, d = synth.synthetic_batch(B, H, W, config_id=config_id)
    r, d = torch.from_numpy(r), torch.from_numpy(d)
    with torch.no_grad():
        y_r, y_d = net.g_a(r, d)
        out = net.compress(r, d)
        dec = net.decompress(out["r_strings"], out["d_strings"], out["shape"])
        fw = net(r, d)  # eval-mode forward (inherited models/elic_united.py:234-263 over the R2D slice coder)
        # the Bi-CEE stage alone (inherited compress_united / decompress_united, elic_united.py:350-401,543-578)
        lat = [torch.from_numpy(a) for a in synth.synthetic_latents(1, 8, 12, 320, 6)]
        cu = net.compress_united(lat[0], lat[1], lat[2], lat[3])
        du = net.decompress_united(cu[0][0], lat[1], cu[1][0], lat[3])
    g = {"B": B, "H": H, "W": W, "config_id": config_id, "shape": np.array(tuple(out["shape"]), np.int32),
         "r_y": np.frombuffer(out["r_strings"][0][0], np.uint8), "d_y": np.frombuffer(out["d_strings"][0][0], np.uint8),
         "r_z0": np.frombuffer(out["r_strings"][1][0], np.uint8), "d_z0": np.frombuffer(out["d_strings"][1][0], np.uint8),
         "y_r": y_r.numpy(), "y_d": y_d.numpy(),
         "fw_xhat_r_sub": fw["x_hat"]["r"][:, :, ::4, ::4].numpy(), "fw_xhat_d_sub": fw["x_hat"]["d"][:, :, ::4, ::4].numpy(),
         "lik_y_r": fw["r_likelihoods"]["y"].numpy(), "lik_y_d": fw["d_likelihoods"]["y"].numpy(),
         "lik_z_r": fw["r_likelihoods"]["z"].numpy(), "lik_z_d": fw["d_likelihoods"]["z"].numpy(),
         "cu_seed": 6, "cu_r_y": np.frombuffer(cu[0][0], np.uint8), "cu_d_y": np.frombuffer(cu[1][0], np.uint8),
         "cu_yhat_r": du[0].numpy(), "cu_yhat_d": du[1].numpy(),
         "xhat_r_sub": dec["x_hat"]["r"][:, :, ::4, ::4].numpy(), "xhat_d_sub": dec["x_hat"]["d"][:, :, ::4, ::4].numpy(),
         "psnr": np.array([-10 * np.log10(torch.mean((dec["x_hat"]["r"] - r) ** 2).item()),
                           -10 * np.log10(torch.mean((dec["x_hat"]["d"] - d) ** 2).item())], np.float64)}
    np.savez_compressed(os.path.join(HERE, f"r2d_{name}.npz"), **g)
    print("r2d", name, len(out["r_strings"][0][0]), len(out["d_strings"][0][0]), g["psnr"])
    return g


def main():
    ELIC, model_config, ext = rl.load_reference()
    import rgbd_amd  # noqa: F401
    from rgbd_amd import synth

    if "--only-elic-fw" in sys.argv:  # eval-mode forward() of the single-modal ELIC (models/elic.py:60-161)
        net = ext["ELIC"](config=model_config(), channel=3).eval()
        net.load_state_dict(synth.synthetic_state_dict(0, model="ELIC"))
        assert net.update(force=True)
        r, _ = synth.synthetic_batch(2, 128, 192, config_id=11)
        with torch.no_grad():
            fw = net(torch.from_numpy(r))
        g = {"B": 2, "H": 128, "W": 192, "config_id": 11, "x_hat": fw["x_hat"].numpy(),
             "lik_y": fw["likelihoods"]["y_likelihoods"].numpy(), "lik_z": fw["likelihoods"]["z_likelihoods"].numpy()}
        np.savez_compressed(os.path.join(HERE, "elic_fw_b2_128x192.npz"), **g)
        print("elic forward", g["x_hat"].shape, float(g["lik_y"].mean()), float(g["lik_z"].mean()))
        return
    if "--only-r2d" in sys.argv:  # refresh one fixture without touching the others
        r2d_case(model_config, synth, "128x192", 1, 128, 192, 4)
        return
    if "--only-hr" in sys.argv:  # the high_rate weights (98 % of the symbols on CDF rows of 300 ... 3000 entries): wide rows end to end
        net = ELIC(config=model_config(), channel=4).eval()
        net.load_state_dict(synth.synthetic_state_dict(0, recipe="high_rate"))
        assert net.update(force=True)
        model_case(net, synth, "i_128x192_hr", 1, 128, 192, 41, False)
        return
    if "--only-r2d-heldout" in sys.argv:  # round 5: a held-out ELIC_united_R2D case (new size, new seed)
        r2d_case(model_config, synth, "o_192x256_s9", 1, 192, 256, 17, seed=9)
        return
    if "--only-single-heldout" in sys.argv:  # round 5: a held-out single-modal case (new size, new seed; see make_margins.py j_ / k_ / l_ / m_)
        elic_single_case(ext, model_config, synth, "n_192x256_s8", 1, 192, 256, 16, seed=8)
        return
    if "--only-e" in sys.argv:  # the bench's image shape (480x640 -> 512x640) with the trained-like weights (~3.6 bpp)
        net = ELIC(config=model_config(), channel=4).eval()
        net.load_state_dict(synth.synthetic_state_dict(0, recipe="trained_like"))
        assert net.update(force=True)
        model_case(net, synth, "e_480x640_tl", 1, 480, 640, 3, False)
        return

    net = ELIC(config=model_config(), channel=4).eval()
    sd = synth.synthetic_state_dict(0)
    net.load_state_dict(sd)
    assert net.update(force=True)

    kat = coder_kats(net, ext)
    kat.update(eb_tables(net))
    np.savez_compressed(os.path.join(HERE, "coder_kat.npz"), **kat)
    print("coder KATs:", len(kat["tiny_stream"]), len(kat["b2_stream"]), sha(kat["gc_cdf"].tobytes()))

    cases = [("a_128x192", 1, 128, 192, 9, True), ("b_100x150", 1, 100, 150, 8, False),
             ("c_b2_128x128", 2, 128, 128, 7, False), ("d_256x256", 1, 256, 256, 2, False)]
    summary = {}
    for name, B, H, W, cid, full in cases:
        g = model_case(net, synth, name, B, H, W, cid, full)
        summary[name] = {"B": B, "H": H, "W": W, "config_id": cid, "padded": g["padded"].tolist(),
                         "shape": g["shape"].tolist(), "psnr": g["psnr"].tolist(),
                         "bpp": g["bpp"].tolist() if "bpp" in g else None,
                         "y_len": [int(g["r_y"].shape[0]), int(g["d_y"].shape[0])]}
    for name, B, h, w, seed in (("c4_16x16", 1, 16, 16, 4), ("c4_b2_8x12", 2, 8, 12, 5)):
        g = bicee_case(net, synth, name, B, h, w, seed)
        summary["bicee_" + name] = {"B": B, "h": h, "w": w, "seed": seed,
                                    "y_len": [int(g["r_y"].shape[0]), int(g["d_y"].shape[0])]}
    g = elic_single_case(ext, model_config, synth, "c1_256x256", 1, 256, 256, 1)
    summary["elic_c1_256x256"] = {"B": 1, "H": 256, "W": 256, "config_id": 1, "y_len": int(g["y_stream"].shape[0]),
                                  "psnr": g["psnr"].tolist()}
    g = stf_case(model_config, synth, "c5_256x256", 1, 256, 256, 5)
    summary["stf_c5_256x256"] = {"B": 1, "H": 256, "W": 256, "config_id": 5, "psnr": g["psnr"].tolist(),
                                 "y_len": [int(g["r_y"].shape[0]), int(g["d_y"].shape[0])]}
    g = r2d_case(model_config, synth, "128x192", 1, 128, 192, 4)
    summary["r2d_128x192"] = {"B": 1, "H": 128, "W": 192, "config_id": 4, "psnr": g["psnr"].tolist(),
                              "y_len": [int(g["r_y"].shape[0]), int(g["d_y"].shape[0])]}
    with open(os.path.join(HERE, "harness.json"), "w") as f:
        json.dump({"weights_seed": 0, "torch": torch.__version__, "cases": summary}, f, indent=1)


if __name__ == "__main__":
    main()
