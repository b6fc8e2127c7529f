#!/usr/bin/env python3
"""fp64 adjudication of the near-boundary decisions.

Every golden whose streams are not identical on the GPU differs from the reference at a value the reference itself
decided within one ulp ... 5.5e-5 of a rounding / table boundary (tests/golden/make_margins.py).  "Within fp32 noise" does
not say which side is nearer the exact value.  This script evaluates the SAME arithmetic in float64 -- oracle/elic_oracle.py
with the fp32 weights and inputs cast to double (exact), every convolution, pooling, sigmoid, LayerNorm ... in double --
and stores the float64 value of every decision the reference took near a boundary:

    f64_zx_{r,d}        z - median at every z position
    f64_nb_x_{r,d}      y - mean   at the positions margins_<case>.npz lists as near-boundary (nb_pos_*)
    f64_nb_s_{r,d}      scale      at the same positions

The y path is TEACHER-FORCED with the reference's own symbols (y_hat = ref_sym + mean64, z_hat = ref_zsym + median): up
to the GPU's first flip that is exactly the context under which the reference and the GPU both decided, and the reference
is compared with fp64 under its own decisions everywhere.  tests/test_gpu_parity_pinned.py then reports, per golden, how
many of those decisions the GPU / the reference take like fp64 and, at the first flip, |gpu - fp64|, |reference - fp64|
and the side of the boundary fp64 falls on.

Needs neither the reference nor a GPU (the oracle is pinned to the reference bit for bit in fp32: tests/test_oracle_*.py):

    python tests/golden/make_fp64.py [case ...]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import elic_oracle as eo  # noqa: E402

SCALE_BOUND = 0.11


def to64(sd):
    return {k: (v.detach().to(torch.float64) if v.is_floating_point() else v) for k, v in sd.items()}


def decisions(x64, s64, table32):
    """The integer decisions the float64 values imply: round-half-even of y - mean, and the scale-table row
    (entropy_models.py:561-568: #{i < 63 : table_i < max(s, 0.11)}, the table being the model's fp32 constants)."""
    sym = np.rint(x64).astype(np.int64)
    idx = np.searchsorted(table32[:-1].astype(np.float64), np.maximum(s64, np.float64(np.float32(SCALE_BOUND))), side="left")
    return sym, idx.astype(np.int64)


class Forced:
    """Per modality: the reference's symbols in stream order, consumed part by part; collects x64 / s64 in that order."""

    def __init__(self, mg, tags):
        self.sym = {t: mg[f"ref_sym_{t}"].astype(np.int64) for t in tags}
        self.pos = {t: 0 for t in tags}
        self.x = {t: [] for t in tags}
        self.s = {t: [] for t in tags}

    def part(self, tag, y_sq, m_sq, s_sq):
        n = m_sq.numel()
        ref = torch.from_numpy(self.sym[tag][self.pos[tag]:self.pos[tag] + n]).reshape(m_sq.shape)
        self.pos[tag] += n
        self.x[tag].append((y_sq - m_sq).reshape(-1).numpy().copy())
        self.s[tag].append(s_sq.reshape(-1).numpy().copy())
        return ref.to(torch.float64) + m_sq

    def pack(self, mg, table32, out):
        for t in self.sym:
            assert self.pos[t] == len(self.sym[t]), (t, self.pos[t], len(self.sym[t]))
            x, s = np.concatenate(self.x[t]), np.concatenate(self.s[t])
            nb = mg[f"nb_pos_{t}"]
            out[f"f64_nb_x_{t}"], out[f"f64_nb_s_{t}"] = x[nb], s[nb]
            # how the reference's own decisions compare with fp64 under the reference's context (all positions)
            sym64, idx64 = decisions(x, s, table32)
            ref_idx = mg[f"ref_idx_{t}"].astype(np.int64)
            out[f"ref_vs_f64_round_differ_{t}"] = np.nonzero(sym64 != self.sym[t])[0].astype(np.int32)
            out[f"ref_vs_f64_index_differ_{t}"] = np.nonzero(idx64 != ref_idx)[0].astype(np.int32)


def bicee64(sd, slice_ch, y, hyp, forced, r2d=False):
    """elic_united.py:265-348 in float64 with the symbols forced (oracle OracleCodec._slice without the integer stage)."""
    yhat = {"r": [], "d": []}
    for i, C in enumerate(slice_ch):
        c0 = sum(slice_ch[:i])
        ctx0 = [hyp["r"], hyp["d"]]
        if i:
            ctx0 = ctx0 + [eo._channel_context(sd, f"rgb_channel_context.{i}", torch.cat(yhat["r"], dim=1)),
                           eo._channel_context(sd, f"depth_channel_context.{i}", torch.cat(yhat["d"], dim=1))]

        def part(tag, mod, anchor, ctx):
            fam = f"{mod}_entropy_parameters_{'anchor' if anchor else 'nonanchor'}.{i}"
            scales, means = eo._entropy_params(sd, fam, torch.cat(ctx, dim=1)).chunk(2, 1)
            s_sq, m_sq = eo.pack(scales, anchor), eo.pack(means, anchor)
            y_sq = eo.pack(y[tag][:, c0:c0 + C], anchor)
            return eo.unpack(forced.part(tag, y_sq, m_sq, s_sq), anchor)

        ra = part("r", "rgb", True, ctx0)
        r_loc = eo._conv(sd, f"rgb_local_context.{i}", ra)
        da = part("d", "depth", True, [r_loc] + ctx0)
        d_loc = eo._conv(sd, f"depth_local_context.{i}", da)
        rn = part("r", "rgb", False, [r_loc, d_loc] + ctx0)
        r_hat = rn + ra
        r_loc2 = eo._conv(sd, f"rgb_local_context_anchor_with_nonanchor.{i}", r_hat)
        dn = part("d", "depth", False, [r_loc2, d_loc] + ctx0)
        yhat["r"].append(r_hat)
        yhat["d"].append(dn + da)


def z_stage(sd, z, mg, out):
    """z - median in float64 at every position; z_hat from the REFERENCE's z symbols."""
    zhat = {}
    for tag, mod in (("r", "rgb"), ("d", "depth")):
        med = sd[f"{mod}_entropy_bottleneck.quantiles"][:, :, 1:2].reshape(1, -1, 1, 1)
        zx = (z[tag] - med).reshape(-1).numpy()
        out[f"f64_zx_{tag}"] = zx
        ref = torch.from_numpy(mg[f"ref_zsym_{tag}"].astype(np.int64)).reshape(z[tag].shape)
        zhat[tag] = ref.to(torch.float64) + med
    return zhat


def finish(name, mg, out):
    path = os.path.join(HERE, f"fp64_{name}.npz")
    np.savez_compressed(path, **out)
    msg = {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()}
    print("fp64", name, msg, os.path.getsize(path))


@torch.no_grad()
def united_case(name, B, H, W, cid, seed=0, recipe=None, model="ELIC_united"):
    from rgbd_amd import synth

    mg = np.load(os.path.join(HERE, f"margins_{name}.npz"))
    kw = {"model": model} if model != "ELIC_united" else {}
    sd32 = synth.synthetic_state_dict(seed, **kw) if recipe is None else synth.synthetic_state_dict(seed, recipe=recipe, **kw)
    sd = to64(sd32)
    table32 = eo.scale_table().numpy().astype(np.float32)
    r, d = synth.synthetic_batch(B, H, W, config_id=cid)
    rp = eo.pad_replicate0(torch.from_numpy(r)).to(torch.float64)
    dp = eo.pad_replicate0(torch.from_numpy(d)).to(torch.float64)
    stf = model == "STF_united"
    slice_ch = [24, 24, 48, 96, 192] if stf else [16, 16, 32, 64, 192]
    y_r, y_d = (eo.g_a_stf if stf else eo.g_a)(sd, rp, dp)
    z_r, z_d = eo.h_a(sd, y_r, y_d)
    out = {}
    zhat = z_stage(sd, {"r": z_r, "d": z_d}, mg, out)
    hyp_r, hyp_d = eo.h_s(sd, zhat["r"], zhat["d"])
    forced = Forced(mg, ("r", "d"))
    bicee64(sd, slice_ch, {"r": y_r, "d": y_d}, {"r": hyp_r, "d": hyp_d}, forced)
    forced.pack(mg, table32, out)
    finish(name, mg, out)


@torch.no_grad()
def bicee_case(name, B, h, w, seed):
    from rgbd_amd import synth

    mg = np.load(os.path.join(HERE, f"margins_bicee_{name}.npz"))
    sd = to64(synth.synthetic_state_dict(0))
    yr, hr, yd, hd = [torch.from_numpy(a).to(torch.float64) for a in synth.synthetic_latents(B, h, w, 320, seed)]
    forced = Forced(mg, ("r", "d"))
    bicee64(sd, [16, 16, 32, 64, 192], {"r": yr, "d": yd}, {"r": hr, "d": hd}, forced)
    out = {}
    forced.pack(mg, eo.scale_table().numpy().astype(np.float32), out)
    finish("bicee_" + name, mg, out)


@torch.no_grad()
def single_case(name, H, W, cid):
    """models/elic.py:161-253 in float64, symbols forced (oracle OracleCodecSingle._slices)."""
    from rgbd_amd import synth

    mg = np.load(os.path.join(HERE, f"margins_elic_{name}.npz"))
    sd = to64(synth.synthetic_state_dict(0, model="ELIC"))
    r, _ = synth.synthetic_batch(1, H, W, config_id=cid)
    x = torch.from_numpy(r).to(torch.float64)
    y = eo._stack1(sd, "g_a.analysis_transform", eo._GA1, x)
    t = torch.relu(eo._conv(sd, "h_a.reduction.0", y))
    t = torch.relu(eo._conv(sd, "h_a.reduction.2", t, stride=2))
    z = eo._conv(sd, "h_a.reduction.4", t, stride=2)
    med = sd["entropy_bottleneck.quantiles"][:, :, 1:2].reshape(1, -1, 1, 1)
    out = {"f64_zx_r": (z - med).reshape(-1).numpy()}
    zhat = torch.from_numpy(mg["ref_zsym_r"].astype(np.int64)).reshape(z.shape).to(torch.float64) + med
    t = torch.relu(eo._deconv(sd, "h_s.increase.0", zhat, stride=2))
    t = torch.relu(eo._deconv(sd, "h_s.increase.2", t, stride=2))
    hyper = eo._deconv(sd, "h_s.increase.4", t, stride=1)
    forced = Forced(mg, ("r",))
    yhat, c0 = [], 0
    for i, c in enumerate([16, 16, 32, 64, 192]):
        ctx = ([eo._channel_context(sd, f"channel_context.{i}", torch.cat(yhat, dim=1))] if i else []) + [hyper]

        def part(anchor, cx):
            fam = f"entropy_parameters_{'anchor' if anchor else 'nonanchor'}.{i}"
            scales, means = eo._entropy_params1(sd, fam, torch.cat(cx, dim=1)).chunk(2, 1)
            s_sq, m_sq = eo.pack(scales, anchor), eo.pack(means, anchor)
            return eo.unpack(forced.part("r", eo.pack(y[:, c0:c0 + c], anchor), m_sq, s_sq), anchor)

        a = part(True, ctx)
        loc = eo._conv(sd, f"local_context.{i}", a)
        n = part(False, [loc] + ctx)
        yhat.append(n + a)
        c0 += c
    forced.pack(mg, eo.scale_table().numpy().astype(np.float32), out)
    finish("elic_" + name, mg, out)


def main():
    torch.set_default_dtype(torch.float64)  # constants the oracle creates on the fly (bounds, masks) follow
    torch.set_num_threads(8)
    import rgbd_amd  # noqa: F401

    want = set(sys.argv[1:])

    def on(n):
        return not want or n in want

    if on("d_256x256"):
        united_case("d_256x256", 1, 256, 256, 2)
    if on("g_256x256_s1"):
        united_case("g_256x256_s1", 1, 256, 256, 2, seed=1)
    if on("h_256x256_s2"):
        united_case("h_256x256_s2", 1, 256, 256, 2, seed=2)
    if on("bicee_c4_b2_8x12"):
        bicee_case("c4_b2_8x12", 2, 8, 12, 5)
    if on("elic_c1_256x256"):
        single_case("c1_256x256", 256, 256, 1)
    if on("stf_c5_256x256"):
        united_case("stf_c5_256x256", 1, 256, 256, 5, model="STF_united")
    if on("e_480x640_tl"):
        united_case("e_480x640_tl", 1, 480, 640, 3, recipe="trained_like")
    if on("f_480x640_stress"):
        united_case("f_480x640_stress", 1, 480, 640, 3)
    if on("i_128x192_hr"):
        united_case("i_128x192_hr", 1, 128, 192, 41, recipe="high_rate")


if __name__ == "__main__":
    main()
