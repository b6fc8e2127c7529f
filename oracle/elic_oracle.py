"""CPU restatement of the reference's ELIC_united compress()/decompress() path.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this; the product package never does.

What it is: the reference's module graph written functionally over a flat state_dict, evaluated with
PyTorch *CPU* eager kernels (the same third-party arithmetic the reference itself runs on: SURVEY.md
§8c last row), plus the plain-C coder of oracle/rans_oracle.c.  Every function cites the reference
lines it follows (paths relative to /root/reference).

Parity status: PINNED in the survey container -- tests/test_oracle_model.py checks that this file
reproduces, bit for bit, the streams / tables / reconstructions that the unmodified reference produced
for the committed golden inputs (tests/golden/make_golden.py ran the reference itself).
"""
import math
import time
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import coder

SCALE_MIN, SCALE_MAX, SCALE_LEVELS = 0.11, 256.0, 64
SCALE_BOUND = 0.11
TAIL_MASS = 1e-9


# --------------------------------------------------------------------------------------------------
# tables (one-off): utils/moduleFunc.py:11-12, entropy_models.py:166-172, 320-360, 489-532
# --------------------------------------------------------------------------------------------------
def scale_table() -> torch.Tensor:
    return torch.exp(torch.linspace(math.log(SCALE_MIN), math.log(SCALE_MAX), SCALE_LEVELS))


def _rows_to_cdf(pmf: torch.Tensor, tail: torch.Tensor, lengths: torch.Tensor, max_len: int) -> np.ndarray:
    # entropy_models.py:166-172 -> ops.cpp:24-81 per row
    out = np.zeros((pmf.shape[0], max_len + 2), dtype=np.int32)
    for i in range(pmf.shape[0]):
        row = torch.cat((pmf[i, : int(lengths[i])], tail[i]), dim=0)
        q = coder.pmf_to_quantized_cdf(row.numpy(), 16)
        out[i, : q.shape[0]] = q.astype(np.int64).astype(np.int32)
    return out


def gaussian_tables(table: Optional[torch.Tensor] = None) -> coder.Tables:
    """GaussianConditional.update (entropy_models.py:511-532)."""
    import scipy.stats

    table = scale_table() if table is None else table
    multiplier = -scipy.stats.norm.ppf(TAIL_MASS / 2)
    center = torch.ceil(table * multiplier).int()
    length = 2 * center + 1
    max_len = int(torch.max(length).item())
    samples = torch.abs(torch.arange(max_len).int() - center[:, None]).float()
    sc = table.unsqueeze(1).float()

    def phi(v):  # entropy_models.py:489-494
        return 0.5 * torch.erfc(float(-(2**-0.5)) * v)

    upper = phi((0.5 - samples) / sc)
    lower = phi((-0.5 - samples) / sc)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    cdf = _rows_to_cdf(pmf, tail, length, max_len)
    return coder.Tables(cdf, (length + 2).numpy(), (-center).numpy())


def _eb_logits(sd, prefix: str, v: torch.Tensor) -> torch.Tensor:
    # entropy_models.py:369-388 (filters = (3,3,3,3))
    logits = v
    for i in range(5):
        logits = torch.matmul(F.softplus(sd[f"{prefix}._matrix{i}"]), logits)
        logits = logits + sd[f"{prefix}._bias{i}"]
        if i < 4:
            logits = logits + torch.tanh(sd[f"{prefix}._factor{i}"]) * torch.tanh(logits)
    return logits


def bottleneck_tables(sd, prefix: str) -> coder.Tables:
    """EntropyBottleneck.update (entropy_models.py:320-360)."""
    q = sd[f"{prefix}.quantiles"]
    med = q[:, 0, 1]
    minima = torch.clamp(torch.ceil(med - q[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(q[:, 0, 2] - med).int(), min=0)
    start = med - minima
    length = maxima + minima + 1
    max_len = int(length.max())
    samples = torch.arange(max_len)[None, :] + start[:, None, None]
    lower = _eb_logits(sd, prefix, samples - 0.5)
    upper = _eb_logits(sd, prefix, samples + 0.5)
    sign = -torch.sign(lower + upper)
    pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
    tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
    cdf = _rows_to_cdf(pmf, tail, length, max_len)
    return coder.Tables(cdf, (length + 2).numpy(), (-minima).numpy())


# --------------------------------------------------------------------------------------------------
# blocks: modules/layers/conv.py, res_blk.py, compressai/layers/layers.py, modules/transform/attention.py
# --------------------------------------------------------------------------------------------------
def _conv(sd, name, x, stride=1, pad=None):
    w = sd[name + ".weight"]
    if pad is None:
        pad = w.shape[-1] // 2
    return F.conv2d(x, w, sd[name + ".bias"], stride=stride, padding=pad)


def _deconv(sd, name, x, stride):
    w = sd[name + ".weight"]  # conv.py:16-24
    return F.conv_transpose2d(x, w, sd[name + ".bias"], stride=stride, padding=w.shape[-1] // 2,
                              output_padding=stride - 1)


def _bottleneck(sd, p, x):  # res_blk.py:7-27
    t = torch.relu(_conv(sd, p + ".branch.0", x))
    t = torch.relu(_conv(sd, p + ".branch.2", t))
    t = _conv(sd, p + ".branch.4", t)
    idn = _conv(sd, p + ".skip", x) if (p + ".skip.weight") in sd else x
    return t + idn


def _res_unit(sd, p, x):  # layers.py:177-196
    t = torch.relu(_conv(sd, p + ".conv.0", x))
    t = torch.relu(_conv(sd, p + ".conv.2", t))
    t = _conv(sd, p + ".conv.4", t)
    return torch.relu(t + x)


def _attention(sd, p, x):  # layers.py:198-213
    a = x
    for u in range(3):
        a = _res_unit(sd, f"{p}.conv_a.{u}", a)
    b = x
    for u in range(3):
        b = _res_unit(sd, f"{p}.conv_b.{u}", b)
    b = _conv(sd, p + ".conv_b.3", b)
    return a * torch.sigmoid(b) + x


def _esa(sd, p, x):  # attention.py:84-97
    c1_ = _conv(sd, p + ".conv1", x)
    c1 = _conv(sd, p + ".conv2", c1_, stride=2, pad=0)
    v = F.max_pool2d(c1, kernel_size=7, stride=3)
    v = torch.relu(_conv(sd, p + ".conv_max", v))
    c3 = torch.relu(_conv(sd, p + ".conv3", v))
    c3 = _conv(sd, p + ".conv3_", c3)
    c3 = F.interpolate(c3, (x.size(2), x.size(3)), mode="bilinear", align_corners=False)
    cf = _conv(sd, p + ".conv_f", c1_)
    m = torch.sigmoid(_conv(sd, p + ".conv4", c3 + cf))
    return x * m


def _bi_spf(sd, p, rgb, depth):  # attention.py:35-48
    rf = torch.relu(_conv(sd, p + ".r_ext", rgb))
    df = torch.relu(_conv(sd, p + ".d_ext", depth))
    r = _esa(sd, p + ".r_esa", torch.cat((rf, df), dim=-3))
    d = _esa(sd, p + ".d_esa", torch.cat((df, rf), dim=-3))
    return r, d


def _se(sd, p, x):  # attention.py:63-67
    b, c = x.shape[:2]
    y = F.adaptive_avg_pool2d(x, 1).view(b, c)
    y = torch.sigmoid(F.linear(torch.relu(F.linear(y, sd[p + ".fc.0.weight"])), sd[p + ".fc.2.weight"]))
    return x * y.view(b, c, 1, 1).expand_as(x)


# --------------------------------------------------------------------------------------------------
# transforms
# --------------------------------------------------------------------------------------------------
_GA = ["conv", "rb", "rb", "rb", "spf", "conv", "rb", "rb", "rb", "attn", "spf", "conv", "rb", "rb", "rb", "spf",
       "conv", "attn"]
_GS = ["attn", "deconv", "spf", "rb", "rb", "rb", "deconv", "attn", "spf", "rb", "rb", "rb", "deconv", "spf", "rb",
       "rb", "rb", "deconv"]


def _stack(sd, root, kinds, rgb, depth):
    # analysis.py:161-174 / synthesis.py:171-184: the two 18-stage stacks advance in lock step;
    # the fusion stage appends each modality's gated features to its own stream
    pr, pd = f"{root}.rgb_{root_kind(root)}_transform", f"{root}.depth_{root_kind(root)}_transform"
    for i, k in enumerate(kinds):
        if k == "spf":
            fr, fd = _bi_spf(sd, f"{pr}.{i}", rgb, depth)
            rgb = torch.cat((rgb, fr), dim=-3)
            depth = torch.cat((depth, fd), dim=-3)
            continue
        outs = []
        for p, x in ((pr, rgb), (pd, depth)):
            n = f"{p}.{i}"
            if k == "conv":
                outs.append(_conv(sd, n, x, stride=2))
            elif k == "deconv":
                outs.append(_deconv(sd, n, x, stride=2))
            elif k == "rb":
                outs.append(_bottleneck(sd, n, x))
            else:
                outs.append(_attention(sd, n, x))
        rgb, depth = outs
    return rgb, depth


def root_kind(root):
    return "analysis" if root == "g_a" else "synthesis"


def g_a(sd, rgb, depth):
    return _stack(sd, "g_a", _GA, rgb, depth)


def g_s(sd, rgb, depth):
    return _stack(sd, "g_s", _GS, rgb, depth)


def h_a(sd, rgb_y, depth_y):  # analysis.py:231-242
    outs = []
    for mod, x in (("rgb", rgb_y), ("depth", depth_y)):
        p = f"h_a.{mod}_reduction"
        t = torch.relu(_conv(sd, p + ".0", x))
        t = torch.relu(_conv(sd, p + ".2", t, stride=2))
        outs.append(_conv(sd, p + ".4", t, stride=2))
    return outs


def _hs_block(sd, p, own, other, last):  # synthesis.py:345-362
    f = _se(sd, p + ".se", torch.cat((own, other), dim=-3))
    f = _deconv(sd, p + ".deconv", f, stride=1 if last else 2)
    return f if last else F.leaky_relu(f, 0.01)


def h_s(sd, rgb_z, depth_z):  # synthesis.py:316-323
    r1 = _hs_block(sd, "h_s.r_h_s1", rgb_z, depth_z, False)
    d1 = _hs_block(sd, "h_s.d_h_s1", depth_z, rgb_z, False)
    r2 = _hs_block(sd, "h_s.r_h_s2", r1, d1, False)
    d2 = _hs_block(sd, "h_s.d_h_s2", d1, r1, False)
    return _hs_block(sd, "h_s.r_h_s3", r2, d2, True), _hs_block(sd, "h_s.d_h_s3", d2, r2, True)


def _entropy_params(sd, p, x):  # entropy.py:69-78
    x = x + _se(sd, p + ".se", x)
    t = torch.relu(_conv(sd, p + ".fusion.0", x))
    t = torch.relu(_conv(sd, p + ".fusion.2", t))
    return _conv(sd, p + ".fusion.4", t)


def _channel_context(sd, p, x):  # context.py:10-30
    t = torch.relu(_conv(sd, p + ".fushion.0", x))
    t = torch.relu(_conv(sd, p + ".fushion.2", t))
    return _conv(sd, p + ".fushion.4", t)


# --------------------------------------------------------------------------------------------------
# checkerboard packing + integer stage: utils/ckbd.py:37-125, entropy_models.py:118-146, 561-568
# --------------------------------------------------------------------------------------------------
def _cols(h, w, anchor: bool) -> torch.Tensor:
    r = torch.arange(h)[:, None]
    k = torch.arange(w // 2)[None, :]
    return 2 * k + ((1 - r % 2) if anchor else (r % 2))  # ckbd.py:51-64


def pack(x, anchor: bool):
    h, w = x.shape[-2:]
    idx = _cols(h, w, anchor).expand(*x.shape[:-1], w // 2)
    return torch.gather(x, -1, idx)


def unpack(xs, anchor: bool):
    h, w2 = xs.shape[-2:]
    out = torch.zeros(*xs.shape[:-1], w2 * 2, dtype=xs.dtype)
    idx = _cols(h, w2 * 2, anchor).expand(*xs.shape[:-1], w2)
    return out.scatter(-1, idx, xs)


def scale_indexes(scales: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    # entropy_models.py:561-568: 63 - #{i<63 : max(s, bound) <= table_i}  ==  #{i<63 : table_i < s'}
    s = torch.max(scales, torch.tensor([SCALE_BOUND]))
    return torch.searchsorted(table[:-1].contiguous(), s.contiguous(), right=False).int()


def quantize_symbols(x: torch.Tensor, means: torch.Tensor) -> torch.Tensor:
    return torch.round(x - means).int()  # entropy_models.py:118-146, round-half-even


# --------------------------------------------------------------------------------------------------
# the codec
# --------------------------------------------------------------------------------------------------
class OracleCodec:
    """Functional mirror of ELIC_united.{update,compress,decompress} (models/elic_united.py:350-586)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], config=None):
        self.sd = {k: v.detach().to(torch.float32) if v.is_floating_point() else v for k, v in state_dict.items()}
        cfg = config or {"N": 192, "M": 320, "slice_ch": [16, 16, 32, 64, 192]}
        self.slice_ch = list(cfg["slice_ch"])
        self.table = scale_table()
        self.gc = None
        self.eb = {}
        self.trace = None  # when a dict, per-part intermediates are recorded
        self.g_a, self.g_s = g_a, g_s  # transform pair (the Swin variant swaps these, models/stf_united.py:641-677)
        self.h_s = h_s
        self.r2d = False  # ELIC_united_R2D context wiring

    def update(self):  # elic_united.py:580-586
        self.gc = gaussian_tables(self.table)
        self.eb = {m: bottleneck_tables(self.sd, f"{m}_entropy_bottleneck") for m in ("rgb", "depth")}
        return True

    # -- z path: entropy_models.py:195-266, 431-446
    def _z_compress(self, mod, z):
        med = self.sd[f"{mod}_entropy_bottleneck.quantiles"][:, :, 1:2].reshape(1, -1, 1, 1)
        sym = torch.round(z - med).int()
        c = z.shape[1]
        idx = torch.arange(c, dtype=torch.int32).view(1, c, 1, 1).expand_as(sym)
        strings = [coder.rans_encode(sym[i].reshape(-1).numpy(), idx[i].reshape(-1).numpy(), self.eb[mod])
                   for i in range(z.shape[0])]
        return strings, sym

    def _z_decompress(self, mod, strings, shape):
        med = self.sd[f"{mod}_entropy_bottleneck.quantiles"][:, :, 1:2].reshape(1, -1, 1, 1)
        c = self.eb[mod].cdf.shape[0]
        idx = torch.arange(c, dtype=torch.int32).view(c, 1, 1).expand(c, shape[0], shape[1]).reshape(-1).numpy()
        outs = []
        for s in strings:
            v = coder.rans_decode(s, idx, self.eb[mod])
            outs.append(torch.from_numpy(v.astype(np.float32)).reshape(c, shape[0], shape[1]))
        return torch.stack(outs) + med

    # -- one slice, four parts in the reference order (elic_united.py:265-348 / 454-541)
    def _slice(self, i, y_r, y_d, hyp_r, hyp_d, yhat_r: List, yhat_d: List, enc, dec):
        sd = self.sd
        ctx0 = [hyp_r, hyp_d]
        if i:
            ctx0 = ctx0 + [_channel_context(sd, f"rgb_channel_context.{i}", torch.cat(yhat_r, dim=1)),
                           _channel_context(sd, f"depth_channel_context.{i}", torch.cat(yhat_d, dim=1))]

        def part(mod, anchor, ctx, y_full):
            fam = f"{mod}_entropy_parameters_{'anchor' if anchor else 'nonanchor'}.{i}"
            params = _entropy_params(sd, fam, torch.cat(ctx, dim=1))
            scales, means = params.chunk(2, 1)
            s_sq, m_sq = pack(scales, anchor), pack(means, anchor)
            idx = scale_indexes(s_sq, self.table)
            if enc is not None:
                sym = quantize_symbols(pack(y_full, anchor), m_sq)
                enc[mod][0].append(sym.reshape(-1).numpy())
                enc[mod][1].append(idx.reshape(-1).numpy())
            else:
                v = dec[mod].decode_stream(idx.reshape(-1).numpy(), self.gc)
                sym = torch.from_numpy(v).reshape(idx.shape)
            if self.trace is not None:
                self.trace.setdefault("parts", []).append(
                    {"slice": i, "mod": mod, "anchor": anchor, "scales": s_sq.clone(), "means": m_sq.clone(),
                     "symbols": sym.clone(), "indexes": idx.clone()})
            return unpack(sym.float() + m_sq, anchor)

        c0 = sum(self.slice_ch[:i])
        c1 = c0 + self.slice_ch[i]
        yr = y_r[:, c0:c1] if y_r is not None else None
        yd = y_d[:, c0:c1] if y_d is not None else None
        # R2D: the RGB nets only see RGB context (hyper_r [, ch_r]); the depth nets see everything
        ctx_r = ([hyp_r] + ctx0[2:3]) if self.r2d else ctx0
        ra = part("rgb", True, ctx_r, yr)
        r_loc = _conv(sd, f"rgb_local_context.{i}", ra)
        da = part("depth", True, [r_loc] + ctx0, yd)
        d_loc = _conv(sd, f"depth_local_context.{i}", da)
        rn = part("rgb", False, ([r_loc] + ctx_r) if self.r2d else ([r_loc, d_loc] + ctx0), yr)
        r_hat = rn + ra
        r_loc2 = _conv(sd, f"rgb_local_context_anchor_with_nonanchor.{i}", r_hat)
        dn = part("depth", False, [r_loc2, d_loc] + ctx0, yd)
        d_hat = dn + da
        yhat_r.append(r_hat)
        yhat_d.append(d_hat)

    @torch.no_grad()
    def compress(self, rgb: torch.Tensor, depth: torch.Tensor):  # elic_united.py:403-427
        y_r, y_d = self.g_a(self.sd, rgb, depth)
        z_r, z_d = h_a(self.sd, y_r, y_d)
        zs_r, _ = self._z_compress("rgb", z_r)
        zh_r = self._z_decompress("rgb", zs_r, z_r.shape[-2:])
        zs_d, _ = self._z_compress("depth", z_d)
        zh_d = self._z_decompress("depth", zs_d, z_d.shape[-2:])
        hyp_r, hyp_d = self.h_s(self.sd, zh_r, zh_d)
        ys_r, ys_d = self.compress_united(y_r, hyp_r, y_d, hyp_d)
        if self.trace is not None:
            self.trace.update({"y_r": y_r, "y_d": y_d, "z_r": z_r, "z_d": z_d, "zhat_r": zh_r, "zhat_d": zh_d,
                               "hyper_r": hyp_r, "hyper_d": hyp_d})
        return {"r_strings": [ys_r, zs_r], "d_strings": [ys_d, zs_d], "shape": tuple(z_r.shape[-2:])}

    @torch.no_grad()
    def compress_united(self, y_r, hyp_r, y_d, hyp_d):  # elic_united.py:350-401
        enc = {"rgb": ([], []), "depth": ([], [])}
        yhat_r, yhat_d = [], []
        for i in range(len(self.slice_ch)):
            self._slice(i, y_r, y_d, hyp_r, hyp_d, yhat_r, yhat_d, enc, None)
        ys = {}
        for mod in ("rgb", "depth"):
            ys[mod] = coder.rans_encode(np.concatenate(enc[mod][0]), np.concatenate(enc[mod][1]), self.gc)
        if self.trace is not None:
            self.trace.update({"yhat_r": torch.cat(yhat_r, 1), "yhat_d": torch.cat(yhat_d, 1)})
        return [ys["rgb"]], [ys["depth"]]

    @torch.no_grad()
    def decompress_united(self, y_string_r, hyp_r, y_string_d, hyp_d):  # elic_united.py:543-578
        dec = {"rgb": coder.RansDecoder(), "depth": coder.RansDecoder()}
        dec["rgb"].set_stream(y_string_r)
        dec["depth"].set_stream(y_string_d)
        yhat_r, yhat_d = [], []
        for i in range(len(self.slice_ch)):
            self._slice(i, None, None, hyp_r, hyp_d, yhat_r, yhat_d, None, dec)
        return torch.cat(yhat_r, 1), torch.cat(yhat_d, 1)

    # -- eval-mode forward: elic_united.py:94-263 (quant == "ste": round in eval), entropy_models.py:391-428, 534-558
    def _eb_forward(self, mod, z):
        p = f"{mod}_entropy_bottleneck"
        med = self.sd[f"{p}.quantiles"][:, :, 1:2]
        x = z.permute(1, 2, 3, 0).contiguous()
        shape = x.size()
        v = x.reshape(x.size(0), 1, -1)
        out = torch.round(v - med) + med
        lower = _eb_logits(self.sd, p, out - 0.5)
        upper = _eb_logits(self.sd, p, out + 0.5)
        sign = -torch.sign(lower + upper)
        lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
        lik = torch.max(lik, torch.tensor([1e-9]))
        return out.reshape(shape).permute(3, 0, 1, 2).contiguous(), lik.reshape(shape).permute(3, 0, 1, 2).contiguous()

    @staticmethod
    def _gc_likelihood(y, scales, means):
        out = torch.round(y - means) + means
        v = torch.abs(out - means)
        s = torch.max(scales, torch.tensor([SCALE_BOUND]))
        c = float(-(2 ** -0.5))
        upper = 0.5 * torch.erfc(c * ((0.5 - v) / s))
        lower = 0.5 * torch.erfc(c * ((-0.5 - v) / s))
        return torch.max(upper - lower, torch.tensor([1e-9]))

    @torch.no_grad()
    def forward(self, rgb, depth):
        sd = self.sd
        y_r, y_d = g_a(sd, rgb, depth)
        z_r, z_d = h_a(sd, y_r, y_d)
        zh_r, zl_r = self._eb_forward("rgb", z_r)
        zh_d, zl_d = self._eb_forward("depth", z_d)
        hyp_r, hyp_d = self.h_s(sd, zh_r, zh_d)
        yhat_r, yhat_d, lik_r, lik_d = [], [], [], []
        for i, C in enumerate(self.slice_ch):
            c0 = sum(self.slice_ch[:i])
            yr, yd = y_r[:, c0:c0 + C], y_d[:, c0:c0 + C]
            ctx0 = [hyp_r, hyp_d]
            if i:
                ctx0 = ctx0 + [_channel_context(sd, f"rgb_channel_context.{i}", torch.cat(yhat_r, dim=1)),
                               _channel_context(sd, f"depth_channel_context.{i}", torch.cat(yhat_d, dim=1))]

            def part(mod, anchor, ctx, y_full):
                fam = f"{mod}_entropy_parameters_{'anchor' if anchor else 'nonanchor'}.{i}"
                scales, means = _entropy_params(sd, fam, torch.cat(ctx, dim=1)).chunk(2, 1)
                sc, mu = unpack(pack(scales, anchor), anchor), unpack(pack(means, anchor), anchor)
                yp = unpack(pack(y_full, anchor), anchor)
                return torch.round(yp - mu) + mu, sc, mu

            ra, s_ra, m_ra = part("rgb", True, ctx0, yr)
            r_loc = _conv(sd, f"rgb_local_context.{i}", ra)
            da, s_da, m_da = part("depth", True, [r_loc] + ctx0, yd)
            d_loc = _conv(sd, f"depth_local_context.{i}", da)
            rn, s_rn, m_rn = part("rgb", False, [r_loc, d_loc] + ctx0, yr)
            r_hat = rn + ra
            r_loc2 = _conv(sd, f"rgb_local_context_anchor_with_nonanchor.{i}", r_hat)
            dn, s_dn, m_dn = part("depth", False, [r_loc2, d_loc] + ctx0, yd)
            yhat_r.append(r_hat)
            yhat_d.append(dn + da)
            lik_r.append(self._gc_likelihood(yr, s_ra + s_rn, m_ra + m_rn))
            lik_d.append(self._gc_likelihood(yd, s_da + s_dn, m_da + m_dn))
        xr, xd = g_s(sd, torch.cat(yhat_r, 1), torch.cat(yhat_d, 1))
        return {"x_hat": {"r": xr, "d": xd}, "r_likelihoods": {"y": torch.cat(lik_r, 1), "z": zl_r},
                "d_likelihoods": {"y": torch.cat(lik_d, 1), "z": zl_d}}

    @torch.no_grad()
    def decompress(self, r_strings, d_strings, shape):  # elic_united.py:429-452
        t0 = time.process_time()
        zh_r = self._z_decompress("rgb", r_strings[1], shape)
        zh_d = self._z_decompress("depth", d_strings[1], shape)
        hyp_r, hyp_d = self.h_s(self.sd, zh_r, zh_d)
        yhat_r, yhat_d = self.decompress_united(r_strings[0][0], hyp_r, d_strings[0][0], hyp_d)
        xr, xd = self.g_s(self.sd, yhat_r, yhat_d)
        return {"x_hat": {"r": xr.clamp_(0, 1), "d": xd.clamp_(0, 1)}, "cost_time": time.process_time() - t0}


# --------------------------------------------------------------------------------------------------
# STF_united (models/stf_united.py): Swin analysis / synthesis transforms on top of ELIC_united's entropy model
# --------------------------------------------------------------------------------------------------
STF_DEPTHS, STF_HEADS, STF_EMBED, STF_WS = (2, 2, 6, 2), (3, 6, 12, 24), 48, 4


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _win_part(x, ws):  # stf_united.py:34-38
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def _win_rev(w, ws, H, W):  # stf_united.py:41-45
    B = int(w.shape[0] / (H * W / ws / ws))
    return w.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def _swin_block(sd, p, x, H, W, shift, heads, mask):  # stf_united.py:48-214
    ws = STF_WS
    B, L, C = x.shape
    shortcut = x
    x = _ln(sd, p + ".norm1", x).view(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = x.shape[1], x.shape[2]
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = _win_part(x, ws).view(-1, ws * ws, C)
    B_, N = xw.shape[0], ws * ws
    qkv = F.linear(xw, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"])
    qkv = qkv.reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4).contiguous()
    q, k, v = qkv[0] * ((C // heads) ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    rpb = sd[p + ".attn.relative_position_bias_table"][sd[p + ".attn.relative_position_index"].view(-1)]
    attn = attn + rpb.view(N, N, -1).permute(2, 0, 1).contiguous().unsqueeze(0)
    if shift > 0:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    xo = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    xo = F.linear(xo, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    x = _win_rev(xo.view(-1, ws, ws, C), ws, Hp, Wp)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    if pad_r > 0 or pad_b > 0:
        x = x[:, :H, :W, :].contiguous()
    x = shortcut + x.view(B, H * W, C)
    t = F.linear(_ln(sd, p + ".norm2", x), sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])
    t = F.linear(F.gelu(t), sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
    return x + t


def _shift_mask(H, W):  # stf_united.py:329-352
    ws, sh = STF_WS, STF_WS // 2
    Hp, Wp = int(np.ceil(H / ws)) * ws, int(np.ceil(W / ws)) * ws
    img = torch.zeros((1, Hp, Wp, 1))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
        for wsl in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = _win_part(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, float(-100.0)).masked_fill(m == 0, float(0.0))


def _basic_layer(sd, p, x, H, W, depth, heads, down):  # stf_united.py:270-366
    mask = _shift_mask(H, W)
    for k in range(depth):
        x = _swin_block(sd, f"{p}.blocks.{k}", x, H, W, 0 if k % 2 == 0 else STF_WS // 2, heads, mask)
    if down == "merge":  # PatchMerging, stf_union.py:217-249
        B, L, C = x.shape
        x = x.view(B, H, W, C)
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], -1)
        x = x.view(B, -1, 4 * C)
        x = F.linear(_ln(sd, p + ".downsample.norm", x), sd[p + ".downsample.reduction.weight"])
        return x, (H + 1) // 2, (W + 1) // 2
    if down == "split":  # PatchSplit, stf_united.py:252-267
        B, L, C = x.shape
        x = F.linear(_ln(sd, p + ".downsample.norm", x), sd[p + ".downsample.reduction.weight"])
        x = F.pixel_shuffle(x.permute(0, 2, 1).contiguous().view(B, 2 * C, H, W), 2)
        return x.permute(0, 2, 3, 1).contiguous().view(B, 4 * L, -1), 2 * H, 2 * W
    return x, H, W


def _stf_stack(sd, root, rgb, depth, Wh, Ww, depths, heads, down):
    B = rgb.shape[0]
    li = 0
    for i in range(4):
        dn = down if i < 3 else None
        rgb, _, _ = _basic_layer(sd, f"{root}.rgb_{root_kind2(root)}_layers.{li}", rgb, Wh, Ww, depths[i], heads[i], dn)
        depth, Wh, Ww = _basic_layer(sd, f"{root}.depth_{root_kind2(root)}_layers.{li}", depth, Wh, Ww, depths[i], heads[i], dn)
        li += 1
        if i < 3:  # Bi-CPT fusion as a residual (stf_united.py:481-489 / 581-589)
            r = rgb.view(B, Wh, Ww, -1).permute(0, 3, 1, 2).contiguous()
            d = depth.view(B, Wh, Ww, -1).permute(0, 3, 1, 2).contiguous()
            rf, df = _bi_spf(sd, f"{root}.rgb_{root_kind2(root)}_layers.{li}", r, d)
            rgb = (r + rf).flatten(2).transpose(1, 2)
            depth = (d + df).flatten(2).transpose(1, 2)
            li += 1
    return rgb, depth, Wh, Ww


def root_kind2(root):
    return "ana" if root == "g_a" else "syn"


def g_a_stf(sd, rgb, depth):  # stf_united.py:462-502
    outs = []
    for mod, x in (("rgb", rgb), ("depth", depth)):
        x = F.conv2d(x, sd[f"g_a.{mod}_patch_embed.proj.weight"], sd[f"g_a.{mod}_patch_embed.proj.bias"], stride=2)
        Wh, Ww = x.shape[2], x.shape[3]
        x = _ln(sd, f"g_a.{mod}_patch_embed.norm", x.flatten(2).transpose(1, 2))
        x = x.transpose(1, 2).view(-1, STF_EMBED, Wh, Ww)
        outs.append(x.flatten(2).transpose(1, 2))
    r, d, Wh, Ww = _stf_stack(sd, "g_a", outs[0], outs[1], Wh, Ww, STF_DEPTHS, STF_HEADS, "merge")
    C = STF_EMBED * 8
    return (r.view(-1, Wh, Ww, C).permute(0, 3, 1, 2).contiguous(), d.view(-1, Wh, Ww, C).permute(0, 3, 1, 2).contiguous())


def g_s_stf(sd, rgb, depth):  # stf_united.py:564-602
    B, C, Wh, Ww = rgb.shape
    r = rgb.permute(0, 2, 3, 1).contiguous().view(-1, Wh * Ww, C)
    d = depth.permute(0, 2, 3, 1).contiguous().view(-1, Wh * Ww, C)
    r, d, Wh, Ww = _stf_stack(sd, "g_s", r, d, Wh, Ww, STF_DEPTHS[::-1], STF_HEADS[::-1], "split")
    outs = []
    for mod, x in (("rgb", r), ("depth", d)):
        x = x.view(-1, Wh, Ww, STF_EMBED).permute(0, 3, 1, 2).contiguous()
        x = F.pixel_shuffle(_conv(sd, f"g_s.{mod}_end_conv.0", x), 2)
        outs.append(_conv(sd, f"g_s.{mod}_end_conv.2", x))
    return outs[0], outs[1]


def oracle_stf(state_dict):
    """OracleCodec for STF_united: ELIC_united's compress/decompress with the Swin transforms and its own widths."""
    c = OracleCodec(state_dict, {"N": 192, "M": 384, "slice_ch": [24, 24, 48, 96, 192]})
    c.g_a, c.g_s = g_a_stf, g_s_stf
    return c


# --------------------------------------------------------------------------------------------------
# ELIC_united_R2D (models/elic_united_R2D.py): RGB is coded on its own, depth is conditioned on RGB
# --------------------------------------------------------------------------------------------------
def _bi_spf_single(sd, p, rgb, depth):  # attention.py:14-32
    rf = torch.relu(_conv(sd, p + ".r_ext", rgb))
    df = torch.relu(_conv(sd, p + ".d_ext", depth))
    return _esa(sd, p + ".d_esa", torch.cat((df, rf), dim=-3))


def _stack_r2d(sd, root, kinds, rgb, depth):  # analysis.py:102-112 / synthesis.py:231-242
    pr, pd = f"{root}.rgb_{root_kind(root)}_transform", f"{root}.depth_{root_kind(root)}_transform"
    for i, k in enumerate(kinds):
        if k == "spf":
            depth = torch.cat((depth, _bi_spf_single(sd, f"{pr}.{i}", rgb, depth)), dim=-3)
            continue
        outs = []
        for p, x in ((pr, rgb), (pd, depth)):
            n = f"{p}.{i}"
            if k == "conv":
                outs.append(_conv(sd, n, x, stride=2))
            elif k == "deconv":
                outs.append(_deconv(sd, n, x, stride=2))
            elif k == "rb":
                outs.append(_bottleneck(sd, n, x))
            else:
                outs.append(_attention(sd, n, x))
        rgb, depth = outs
    return rgb, depth


def g_a_r2d(sd, rgb, depth):
    return _stack_r2d(sd, "g_a", _GA, rgb, depth)


def g_s_r2d(sd, rgb, depth):
    return _stack_r2d(sd, "g_s", _GS, rgb, depth)


def _hs_block_single(sd, p, f, last):  # synthesis.py:364-380
    f = _deconv(sd, p + ".deconv", _se(sd, p + ".se", f), stride=1 if last else 2)
    return f if last else F.leaky_relu(f, 0.01)


def h_s_r2d(sd, rgb_z, depth_z):  # synthesis.py:336-343
    r1 = _hs_block_single(sd, "h_s.r_h_s1", rgb_z, False)
    d1 = _hs_block(sd, "h_s.d_h_s1", depth_z, rgb_z, False)
    r2 = _hs_block_single(sd, "h_s.r_h_s2", r1, False)
    d2 = _hs_block(sd, "h_s.d_h_s2", d1, r1, False)
    return _hs_block_single(sd, "h_s.r_h_s3", r2, True), _hs_block(sd, "h_s.d_h_s3", d2, r2, True)


def oracle_r2d(state_dict, config=None):
    """OracleCodec for ELIC_united_R2D: other transforms, hyper synthesis and context wiring
    (elic_united_R2D.py:149-232 compress_one_slice / :234-326 decompress_one_slice)."""
    c = OracleCodec(state_dict, config)
    c.g_a, c.g_s, c.h_s, c.r2d = g_a_r2d, g_s_r2d, h_s_r2d, True
    return c


# --------------------------------------------------------------------------------------------------
# single-modal ELIC (models/elic.py) -- BASELINE config 1 / SURVEY §8f rank 4
# --------------------------------------------------------------------------------------------------
_GA1 = ["conv", "rb", "rb", "rb", "conv", "rb", "rb", "rb", "attn", "conv", "rb", "rb", "rb", "conv", "attn"]
_GS1 = ["attn", "deconv", "rb", "rb", "rb", "deconv", "attn", "rb", "rb", "rb", "deconv", "rb", "rb", "rb", "deconv"]


def _stack1(sd, prefix, kinds, x):  # analysis.py:29-52 / synthesis.py:32-70
    for i, k in enumerate(kinds):
        n = f"{prefix}.{i}"
        if k == "conv":
            x = _conv(sd, n, x, stride=2)
        elif k == "deconv":
            x = _deconv(sd, n, x, stride=2)
        elif k == "rb":
            x = _bottleneck(sd, n, x)
        else:
            x = _attention(sd, n, x)
    return x


def _entropy_params1(sd, p, x):  # entropy.py:7-29 (three 1x1 convolutions)
    t = torch.relu(_conv(sd, p + ".fusion.0", x))
    t = torch.relu(_conv(sd, p + ".fusion.2", t))
    return _conv(sd, p + ".fusion.4", t)


class OracleCodecSingle:
    """Functional mirror of ELIC.{update,compress,decompress} (models/elic.py:161-351)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], config=None):
        self.sd = {k: v.detach().to(torch.float32) if v.is_floating_point() else v for k, v in state_dict.items()}
        cfg = config or {"N": 192, "M": 320, "slice_ch": [16, 16, 32, 64, 192]}
        self.slice_ch = list(cfg["slice_ch"])
        self.table = scale_table()
        self.gc = None
        self.eb = None
        self.trace = None

    def update(self):  # elic.py:327-332
        self.gc = gaussian_tables(self.table)
        self.eb = bottleneck_tables(self.sd, "entropy_bottleneck")
        return True

    def _median(self):
        return self.sd["entropy_bottleneck.quantiles"][:, :, 1:2].reshape(1, -1, 1, 1)

    def _z_compress(self, z):  # entropy_models.py:195-224, 431-440
        sym = torch.round(z - self._median()).int()
        c = z.shape[1]
        idx = torch.arange(c, dtype=torch.int32).view(1, c, 1, 1).expand_as(sym)
        return [coder.rans_encode(sym[i].reshape(-1).numpy(), idx[i].reshape(-1).numpy(), self.eb)
                for i in range(z.shape[0])]

    def _z_decompress(self, strings, shape):  # entropy_models.py:226-266, 442-446
        c = self.eb.cdf.shape[0]
        idx = torch.arange(c, dtype=torch.int32).view(c, 1, 1).expand(c, shape[0], shape[1]).reshape(-1).numpy()
        outs = [torch.from_numpy(coder.rans_decode(s, idx, self.eb).astype(np.float32)).reshape(c, shape[0], shape[1])
                for s in strings]
        return torch.stack(outs) + self._median()

    def _slices(self, y, hyper, enc, dec):  # elic.py:180-251 / 268-316
        sd, yhat = self.sd, []

        def part(i, anchor, ctx, y_slice):
            fam = f"entropy_parameters_{'anchor' if anchor else 'nonanchor'}.{i}"
            scales, means = _entropy_params1(sd, fam, torch.cat(ctx, dim=1)).chunk(2, 1)
            s_sq, m_sq = pack(scales, anchor), pack(means, anchor)
            idx = scale_indexes(s_sq, self.table)
            if enc is not None:
                sym = quantize_symbols(pack(y_slice, anchor), m_sq)
                enc[0].append(sym.reshape(-1).numpy())
                enc[1].append(idx.reshape(-1).numpy())
            else:
                sym = torch.from_numpy(dec.decode_stream(idx.reshape(-1).numpy(), self.gc)).reshape(idx.shape)
            if self.trace is not None:
                self.trace.setdefault("parts", []).append(
                    {"slice": i, "mod": "rgb", "anchor": anchor, "scales": s_sq.clone(), "means": m_sq.clone(),
                     "symbols": sym.clone(), "indexes": idx.clone()})
            return unpack(sym.float() + m_sq, anchor)

        c0 = 0
        for i, c in enumerate(self.slice_ch):
            ys = y[:, c0:c0 + c] if y is not None else None
            ctx = ([_channel_context(sd, f"channel_context.{i}", torch.cat(yhat, dim=1))] if i else []) + [hyper]
            a = part(i, True, ctx, ys)
            loc = _conv(sd, f"local_context.{i}", a)
            n = part(i, False, [loc] + ctx, ys)
            yhat.append(n + a)
            c0 += c
        return torch.cat(yhat, dim=1)

    @torch.no_grad()
    def compress(self, x: torch.Tensor):  # elic.py:161-253
        y = _stack1(self.sd, "g_a.analysis_transform", _GA1, x)
        t = torch.relu(_conv(self.sd, "h_a.reduction.0", y))
        t = torch.relu(_conv(self.sd, "h_a.reduction.2", t, stride=2))
        z = _conv(self.sd, "h_a.reduction.4", t, stride=2)
        zs = self._z_compress(z)
        zhat = self._z_decompress(zs, z.shape[-2:])
        hyper = self._hyper(zhat)
        enc = ([], [])
        yhat = self._slices(y, hyper, enc, None)
        ys = coder.rans_encode(np.concatenate(enc[0]), np.concatenate(enc[1]), self.gc)
        if self.trace is not None:
            self.trace.update({"y": y, "z": z, "zhat": zhat, "hyper": hyper, "yhat": yhat})
        return {"strings": [[ys], zs], "shape": tuple(z.shape[-2:])}

    def _hyper(self, zhat):  # synthesis.py:276-285
        t = torch.relu(_deconv(self.sd, "h_s.increase.0", zhat, stride=2))
        t = torch.relu(_deconv(self.sd, "h_s.increase.2", t, stride=2))
        return _deconv(self.sd, "h_s.increase.4", t, stride=1)

    @torch.no_grad()
    def forward(self, x: torch.Tensor):
        """Eval-mode forward of the single-modal model (models/elic.py:60-161 with config quant = "ste"): per slice
        y_hat = round(y - mean) + mean on the anchor positions, then on the non-anchor positions with the local context of
        the anchors; Gaussian likelihoods of the merged parameters, factorised-prior likelihoods of round(z - median)."""
        sd = self.sd
        y = _stack1(sd, "g_a.analysis_transform", _GA1, x)
        t = torch.relu(_conv(sd, "h_a.reduction.0", y))
        t = torch.relu(_conv(sd, "h_a.reduction.2", t, stride=2))
        z = _conv(sd, "h_a.reduction.4", t, stride=2)
        # entropy_models.py:369-428 on the quantised z (same arithmetic as OracleCodec._eb_forward)
        p = "entropy_bottleneck"
        med = sd[f"{p}.quantiles"][:, :, 1:2]
        zt = z.permute(1, 2, 3, 0).contiguous()
        shape = zt.size()
        out = torch.round(zt.reshape(zt.size(0), 1, -1) - med) + med
        lower, upper = _eb_logits(sd, p, out - 0.5), _eb_logits(sd, p, out + 0.5)
        sign = -torch.sign(lower + upper)
        lz = torch.max(torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower)), torch.tensor([1e-9]))
        zhat = out.reshape(shape).permute(3, 0, 1, 2).contiguous()
        lz = lz.reshape(shape).permute(3, 0, 1, 2).contiguous()
        hyper = self._hyper(zhat)

        def keep(t_, anchor):  # ckbd_anchor / ckbd_nonanchor: the other checkerboard half becomes zero
            return unpack(pack(t_, anchor), anchor)

        yhat, ly = [], []
        c0 = 0
        for i, c in enumerate(self.slice_ch):
            ys = y[:, c0:c0 + c]
            ctx = ([_channel_context(sd, f"channel_context.{i}", torch.cat(yhat, dim=1))] if i else []) + [hyper]
            sa, ma = _entropy_params1(sd, f"entropy_parameters_anchor.{i}", torch.cat(ctx, dim=1)).chunk(2, 1)
            sa, ma = keep(sa, True), keep(ma, True)
            a = keep(torch.round(keep(ys, True) - ma) + ma, True)  # elic.py:84-85 (ste_round on the anchor half)
            loc = _conv(sd, f"local_context.{i}", a)
            sn, mn = _entropy_params1(sd, f"entropy_parameters_nonanchor.{i}", torch.cat([loc] + ctx, dim=1)).chunk(2, 1)
            sn, mn = keep(sn, False), keep(mn, False)
            n = keep(torch.round(keep(ys, False) - mn) + mn, False)
            ly.append(OracleCodec._gc_likelihood(ys, sa + sn, ma + mn))  # ckbd_merge, then entropy_models.py:534-558
            yhat.append(a + n)
            c0 += c
        yhat = torch.cat(yhat, dim=1)
        return {"x_hat": _stack1(sd, "g_s.synthesis_transform", _GS1, yhat),
                "likelihoods": {"y_likelihoods": torch.cat(ly, dim=1), "z_likelihoods": lz}}

    @torch.no_grad()
    def decompress(self, strings, shape):  # elic.py:255-325
        zhat = self._z_decompress(strings[1], shape)
        hyper = self._hyper(zhat)
        dec = coder.RansDecoder()
        dec.set_stream(strings[0][0])
        yhat = self._slices(None, hyper, None, dec)
        return {"x_hat": _stack1(self.sd, "g_s.synthesis_transform", _GS1, yhat), "y_hat": yhat}


# --------------------------------------------------------------------------------------------------
# harness arithmetic: dataset/utils.py:58-100, utils/IOutils.py:30-88, utils/metrics.py:8-14
# --------------------------------------------------------------------------------------------------
def pad_replicate0(x: torch.Tensor, p: int = 64) -> torch.Tensor:
    H, W = x.shape[-2:]
    ph = (p * (H // p + 1) - H) if H % p else 0
    pw = (p * (W // p + 1) - W) if W % p else 0
    return F.pad(x, (0, pw, 0, ph), mode="replicate")


def container_bytes(H: int, W: int, shape, strings) -> bytes:
    import struct

    out = struct.pack(">2I", H, W) + struct.pack(">3I", shape[0], shape[1], len(strings))
    for lst in strings:
        out += struct.pack(">I", len(lst))
        for s in lst:
            out += struct.pack(">I", len(s)) + bytes(s)
    return out


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    mse = torch.mean((a.clamp(0, 1) - b.clamp(0, 1)) ** 2).item()
    return float(20 * np.log10(1.0) - 10 * np.log10(mse))
