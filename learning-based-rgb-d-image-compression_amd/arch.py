"""Parameter inventory of the ELIC_united hot path.

The reference checkpoint format is the interface: this module enumerates every
state_dict entry of `ELIC_united` (reference: models/elic_united.py:14-86) by name, shape and role, so
that (a) checkpoints written by the reference load here, (b) synthetic weights can be produced for any
entry, and (c) the device-side weight packer knows what each tensor is.

Nothing here is executable network code; it is a table.
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

# config/config.py:5-10 of the reference
DEFAULT_CONFIG = {
    "N": 192,
    "M": 320,
    "slice_num": 5,
    "context_window": 5,
    "slice_ch": [16, 16, 32, 64, 192],
    "quant": "ste",
}


class Config(dict):
    """Attribute dict with the same access pattern as the reference's utils/IOutils.py:14-23."""

    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def model_config() -> Config:
    return Config({k: (list(v) if isinstance(v, list) else v) for k, v in DEFAULT_CONFIG.items()})


@dataclass(frozen=True)
class Entry:
    shape: Tuple[int, ...]
    kind: str  # conv_w | deconv_w | bias | linear_w | eb_matrix | eb_bias | eb_factor | eb_quantiles | buffer
    dtype: str = "float32"
    fan_in: int = 0
    is_param: bool = True


class _Builder:
    def __init__(self):
        self.entries: "OrderedDict[str, Entry]" = OrderedDict()

    def conv(self, name: str, cin: int, cout: int, k: int):
        self.entries[f"{name}.weight"] = Entry((cout, cin, k, k), "conv_w", fan_in=cin * k * k)
        self.entries[f"{name}.bias"] = Entry((cout,), "bias", fan_in=cin * k * k)

    def deconv(self, name: str, cin: int, cout: int, k: int):
        # ConvTranspose2d weight is (Cin, Cout, kH, kW); torch's fan_in for it is size(1)*k*k
        self.entries[f"{name}.weight"] = Entry((cin, cout, k, k), "deconv_w", fan_in=cout * k * k)
        self.entries[f"{name}.bias"] = Entry((cout,), "bias", fan_in=cout * k * k)

    def linear_nobias(self, name: str, cin: int, cout: int):
        self.entries[f"{name}.weight"] = Entry((cout, cin), "linear_w", fan_in=cin)

    # modules/layers/res_blk.py:7-27
    def bottleneck(self, name: str, n: int, out: Optional[int] = None):
        out = n if out is None else out
        self.conv(f"{name}.branch.0", n, n // 2, 1)
        self.conv(f"{name}.branch.2", n // 2, n // 2, 3)
        self.conv(f"{name}.branch.4", n // 2, out, 1)
        if out != n:
            self.conv(f"{name}.skip", n, out, 1)

    # CompressAI/compressai/layers/layers.py:162-213
    def attention(self, name: str, n: int):
        for br in ("conv_a", "conv_b"):
            for u in range(3):
                self.conv(f"{name}.{br}.{u}.conv.0", n, n // 2, 1)
                self.conv(f"{name}.{br}.{u}.conv.2", n // 2, n // 2, 3)
                self.conv(f"{name}.{br}.{u}.conv.4", n // 2, n, 1)
        self.conv(f"{name}.conv_b.3", n, n, 1)

    # modules/transform/attention.py:70-83
    def esa(self, name: str, n: int):
        f = n // 4
        self.conv(f"{name}.conv1", n, f, 1)
        self.conv(f"{name}.conv_f", f, f, 1)
        self.conv(f"{name}.conv_max", f, f, 3)
        self.conv(f"{name}.conv2", f, f, 3)
        self.conv(f"{name}.conv3", f, f, 3)
        self.conv(f"{name}.conv3_", f, f, 3)
        self.conv(f"{name}.conv4", f, n, 1)

    # modules/transform/attention.py:14-48
    def bi_spf(self, name: str, n: int):
        self.conv(f"{name}.r_ext", n, n // 2, 3)
        self.conv(f"{name}.d_ext", n, n // 2, 3)
        self.esa(f"{name}.d_esa", n)
        self.esa(f"{name}.r_esa", n)

    # modules/transform/attention.py:52-61
    def se(self, name: str, c: int, reduction: int = 16):
        self.linear_nobias(f"{name}.fc.0", c, c // reduction)
        self.linear_nobias(f"{name}.fc.2", c // reduction, c)

    # modules/transform/entropy.py:56-67
    def entropy_params(self, name: str, in_dim: int, out_dim: int):
        self.conv(f"{name}.fusion.0", in_dim, in_dim // 6, 1)
        self.conv(f"{name}.fusion.2", in_dim // 6, out_dim * 4 // 3, 3)
        self.conv(f"{name}.fusion.4", out_dim * 4 // 3, out_dim, 5)
        self.se(f"{name}.se", in_dim)

    # modules/transform/context.py:10-19 (attribute really is spelled "fushion")
    def channel_context(self, name: str, in_dim: int, out_dim: int):
        self.conv(f"{name}.fushion.0", in_dim, 224, 5)
        self.conv(f"{name}.fushion.2", 224, 128, 5)
        self.conv(f"{name}.fushion.4", 128, out_dim, 5)

    # CompressAI/compressai/entropy_models/entropy_models.py:282-314
    def entropy_bottleneck(self, name: str, channels: int, filters=(3, 3, 3, 3)):
        f = (1,) + tuple(filters) + (1,)
        for i in range(len(filters) + 1):
            self.entries[f"{name}._matrix{i}"] = Entry((channels, f[i + 1], f[i]), "eb_matrix")
            self.entries[f"{name}._bias{i}"] = Entry((channels, f[i + 1], 1), "eb_bias")
            if i < len(filters):
                self.entries[f"{name}._factor{i}"] = Entry((channels, f[i + 1], 1), "eb_factor")
        self.entries[f"{name}.quantiles"] = Entry((channels, 1, 3), "eb_quantiles")
        self.buffer(f"{name}._offset", (0,), "int32")
        self.buffer(f"{name}._quantized_cdf", (0,), "int32")
        self.buffer(f"{name}._cdf_length", (0,), "int32")
        self.buffer(f"{name}.target", (3,))
        self.buffer(f"{name}.likelihood_lower_bound.bound", (1,))

    # entropy_models.py:462-485
    def gaussian_conditional(self, name: str):
        self.buffer(f"{name}._offset", (0,), "int32")
        self.buffer(f"{name}._quantized_cdf", (0,), "int32")
        self.buffer(f"{name}._cdf_length", (0,), "int32")
        self.buffer(f"{name}.scale_table", (0,))
        self.buffer(f"{name}.scale_bound", (1,))
        self.buffer(f"{name}.likelihood_lower_bound.bound", (1,))
        self.buffer(f"{name}.lower_bound_scale.bound", (1,))

    # nn.Linear / nn.LayerNorm / Swin pieces of models/stf_united.py
    def linear(self, name: str, cin: int, cout: int, bias: bool = True):
        self.entries[f"{name}.weight"] = Entry((cout, cin), "linear_w", fan_in=cin)
        if bias:
            self.entries[f"{name}.bias"] = Entry((cout,), "bias", fan_in=cin)

    def layernorm(self, name: str, c: int):
        self.entries[f"{name}.weight"] = Entry((c,), "ln_w")
        self.entries[f"{name}.bias"] = Entry((c,), "ln_b")

    # stf_united.py:118-214 (window 4x4: 49 relative offsets; relative_position_index is a registered buffer)
    def swin_block(self, name: str, dim: int, heads: int, window: int = 4, mlp_ratio: int = 4):
        self.layernorm(f"{name}.norm1", dim)
        self.entries[f"{name}.attn.relative_position_bias_table"] = Entry(((2 * window - 1) ** 2, heads), "rpb_table")
        self.buffer(f"{name}.attn.relative_position_index", (window * window, window * window), "int64")
        self.linear(f"{name}.attn.qkv", dim, 3 * dim)
        self.linear(f"{name}.attn.proj", dim, dim)
        self.layernorm(f"{name}.norm2", dim)
        self.linear(f"{name}.mlp.fc1", dim, mlp_ratio * dim)
        self.linear(f"{name}.mlp.fc2", mlp_ratio * dim, dim)

    def buffer(self, name: str, shape, dtype: str = "float32"):
        self.entries[name] = Entry(tuple(shape), "buffer", dtype=dtype, is_param=False)


def slice_offsets(slice_ch: List[int]) -> List[int]:
    out, acc = [], 0
    for c in slice_ch:
        out.append(acc)
        acc += c
    return out


def entropy_param_in_dims(M: int, slice_ch: List[int], i: int) -> Dict[str, int]:
    """Input widths of the four EntropyParametersEX nets of slice i (models/elic_united.py:53-78)."""
    c = slice_ch[i]
    base = 4 * M
    if i == 0:
        return {"rgb_anchor": base, "depth_anchor": base + 2 * c, "rgb_nonanchor": base + 4 * c,
                "depth_nonanchor": base + 4 * c}
    return {"rgb_anchor": base + 4 * c, "depth_anchor": base + 6 * c, "rgb_nonanchor": base + 8 * c,
            "depth_nonanchor": base + 8 * c}


def elic_united_entries(config=None) -> "OrderedDict[str, Entry]":
    """Every state_dict entry of ELIC_united, in a stable order."""
    cfg = model_config() if config is None else config
    N, M = int(cfg["N"]), int(cfg["M"])
    slice_ch = list(cfg["slice_ch"])
    b = _Builder()

    # g_a: modules/transform/analysis.py:116-159
    for mod, cin in (("rgb", 3), ("depth", 1)):
        p = f"g_a.{mod}_analysis_transform"
        b.conv(f"{p}.0", cin, N, 5)
        for j in (1, 2, 3):
            b.bottleneck(f"{p}.{j}", N)
        if mod == "rgb":
            b.bi_spf(f"{p}.4", N)
        b.conv(f"{p}.5", 2 * N, N, 5)
        for j in (6, 7, 8):
            b.bottleneck(f"{p}.{j}", N)
        b.attention(f"{p}.9", N)
        if mod == "rgb":
            b.bi_spf(f"{p}.10", N)
        b.conv(f"{p}.11", 2 * N, N, 5)
        for j in (12, 13, 14):
            b.bottleneck(f"{p}.{j}", N)
        if mod == "rgb":
            b.bi_spf(f"{p}.15", N)
        b.conv(f"{p}.16", 2 * N, M, 5)
        b.attention(f"{p}.17", M)

    # g_s: modules/transform/synthesis.py:126-169
    for mod, cout in (("rgb", 3), ("depth", 1)):
        p = f"g_s.{mod}_synthesis_transform"
        b.attention(f"{p}.0", M)
        b.deconv(f"{p}.1", M, N, 5)
        if mod == "rgb":
            b.bi_spf(f"{p}.2", N)
        b.bottleneck(f"{p}.3", 2 * N, N)
        b.bottleneck(f"{p}.4", N)
        b.bottleneck(f"{p}.5", N)
        b.deconv(f"{p}.6", N, N, 5)
        b.attention(f"{p}.7", N)
        if mod == "rgb":
            b.bi_spf(f"{p}.8", N)
        b.bottleneck(f"{p}.9", 2 * N, N)
        b.bottleneck(f"{p}.10", N)
        b.bottleneck(f"{p}.11", N)
        b.deconv(f"{p}.12", N, N, 5)
        if mod == "rgb":
            b.bi_spf(f"{p}.13", N)
        b.bottleneck(f"{p}.14", 2 * N, N)
        b.bottleneck(f"{p}.15", N)
        b.bottleneck(f"{p}.16", N)
        b.deconv(f"{p}.17", N, cout, 5)

    # h_a: analysis.py:231-237
    for mod in ("rgb", "depth"):
        p = f"h_a.{mod}_reduction"
        b.conv(f"{p}.0", M, N, 3)
        b.conv(f"{p}.2", N, N, 5)
        b.conv(f"{p}.4", N, N, 5)

    # h_s: synthesis.py:305-314, 345-354
    for m in ("r", "d"):
        for idx, (cin, cout, k) in enumerate(((2 * N, M, 5), (2 * M, M * 3 // 2, 5), (3 * M, 2 * M, 3)), 1):
            p = f"h_s.{m}_h_s{idx}"
            b.se(f"{p}.se", cin)
            b.deconv(f"{p}.deconv", cin, cout, k)

    # Bi-CEE nets: models/elic_united.py:30-78
    for fam in ("rgb_local_context", "rgb_local_context_anchor_with_nonanchor", "depth_local_context"):
        for i, c in enumerate(slice_ch):
            b.conv(f"{fam}.{i}", c, 2 * c, 5)
    for fam in ("rgb_channel_context", "depth_channel_context"):
        for i in range(1, len(slice_ch)):
            b.channel_context(f"{fam}.{i}", sum(slice_ch[:i]), 2 * slice_ch[i])
    for i, c in enumerate(slice_ch):
        dims = entropy_param_in_dims(M, slice_ch, i)
        b.entropy_params(f"rgb_entropy_parameters_anchor.{i}", dims["rgb_anchor"], 2 * c)
    for i, c in enumerate(slice_ch):
        dims = entropy_param_in_dims(M, slice_ch, i)
        b.entropy_params(f"depth_entropy_parameters_anchor.{i}", dims["depth_anchor"], 2 * c)
    for i, c in enumerate(slice_ch):
        dims = entropy_param_in_dims(M, slice_ch, i)
        b.entropy_params(f"rgb_entropy_parameters_nonanchor.{i}", dims["rgb_nonanchor"], 2 * c)
    for i, c in enumerate(slice_ch):
        dims = entropy_param_in_dims(M, slice_ch, i)
        b.entropy_params(f"depth_entropy_parameters_nonanchor.{i}", dims["depth_nonanchor"], 2 * c)

    b.entropy_bottleneck("rgb_entropy_bottleneck", N)
    b.entropy_bottleneck("depth_entropy_bottleneck", N)
    b.gaussian_conditional("rgb_gaussian_conditional")
    b.gaussian_conditional("depth_gaussian_conditional")
    return b.entries


def elic_entries(config=None, channel: int = 3) -> "OrderedDict[str, Entry]":
    """Every state_dict entry of the single-modal ELIC (reference: models/elic.py:15-57; 409 tensors, 36,932,427
    parameters for channel=3) -- BASELINE config 1 / SURVEY §8f rank 4."""
    cfg = model_config() if config is None else config
    N, M = int(cfg["N"]), int(cfg["M"])
    slice_ch = list(cfg["slice_ch"])
    b = _Builder()
    # modules/transform/analysis.py:29-52
    p = "g_a.analysis_transform"
    b.conv(f"{p}.0", channel, N, 5)
    for j in (1, 2, 3, 5, 6, 7, 10, 11, 12):
        b.bottleneck(f"{p}.{j}", N)
    b.conv(f"{p}.4", N, N, 5)
    b.attention(f"{p}.8", N)
    b.conv(f"{p}.9", N, N, 5)
    b.conv(f"{p}.13", N, M, 5)
    b.attention(f"{p}.14", M)
    # modules/transform/synthesis.py:32-51
    p = "g_s.synthesis_transform"
    b.attention(f"{p}.0", M)
    b.deconv(f"{p}.1", M, N, 5)
    for j in (2, 3, 4, 7, 8, 9, 11, 12, 13):
        b.bottleneck(f"{p}.{j}", N)
    b.deconv(f"{p}.5", N, N, 5)
    b.attention(f"{p}.6", N)
    b.deconv(f"{p}.10", N, N, 5)
    b.deconv(f"{p}.14", N, channel, 5)
    # analysis.py:207-216, synthesis.py:276-285
    b.conv("h_a.reduction.0", M, N, 3)
    b.conv("h_a.reduction.2", N, N, 5)
    b.conv("h_a.reduction.4", N, N, 5)
    b.deconv("h_s.increase.0", N, M, 5)
    b.deconv("h_s.increase.2", M, M * 3 // 2, 5)
    b.deconv("h_s.increase.4", M * 3 // 2, 2 * M, 3)
    # models/elic.py:32-53; EntropyParameters (entropy.py:7-17) is three 1x1 convolutions
    for i, c in enumerate(slice_ch):
        b.conv(f"local_context.{i}", c, 2 * c, 5)
    for i in range(1, len(slice_ch)):
        b.channel_context(f"channel_context.{i}", sum(slice_ch[:i]), 2 * slice_ch[i])
    for fam, extra in (("entropy_parameters_anchor", 0), ("entropy_parameters_nonanchor", 2)):
        for i, c in enumerate(slice_ch):
            in_dim, out = 2 * M + (extra + (2 if i else 0)) * c, 2 * c
            b.conv(f"{fam}.{i}.fusion.0", in_dim, out * 5 // 3, 1)
            b.conv(f"{fam}.{i}.fusion.2", out * 5 // 3, out * 4 // 3, 1)
            b.conv(f"{fam}.{i}.fusion.4", out * 4 // 3, out, 1)
    b.entropy_bottleneck("entropy_bottleneck", N)
    b.gaussian_conditional("gaussian_conditional")
    return b.entries


def r2d_entropy_param_in_dims(M: int, slice_ch: List[int], i: int) -> Dict[str, int]:
    """models/elic_united_R2D.py:47-71: the RGB nets see RGB context only, the depth nets see both."""
    c = slice_ch[i]
    if i == 0:
        return {"rgb_anchor": 2 * M, "depth_anchor": 4 * M + 2 * c, "rgb_nonanchor": 2 * M + 2 * c,
                "depth_nonanchor": 4 * M + 4 * c}
    return {"rgb_anchor": 2 * M + 2 * c, "depth_anchor": 4 * M + 6 * c, "rgb_nonanchor": 2 * M + 4 * c,
            "depth_nonanchor": 4 * M + 8 * c}


def elic_united_r2d_entries(config=None) -> "OrderedDict[str, Entry]":
    """Every state_dict entry of ELIC_united_R2D (reference: models/elic_united_R2D.py:9-71): the one-directional variant
    (RGB is coded on its own, depth is conditioned on RGB) -- SURVEY 8f rank 4."""
    cfg = model_config() if config is None else config
    N, M = int(cfg["N"]), int(cfg["M"])
    slice_ch = list(cfg["slice_ch"])
    b = _Builder()

    def spf_single(name, n):  # modules/transform/attention.py:14-32
        b.conv(f"{name}.r_ext", n, n // 2, 3)
        b.conv(f"{name}.d_ext", n, n // 2, 3)
        b.esa(f"{name}.d_esa", n)

    # analysis.py:56-112: the RGB stream never takes depth features; the depth stream concatenates the fused ones
    for mod, cin, k in (("rgb", 3, 1), ("depth", 1, 2)):
        p = f"g_a.{mod}_analysis_transform"
        b.conv(f"{p}.0", cin, N, 5)
        for j in (1, 2, 3, 6, 7, 8, 12, 13, 14):
            b.bottleneck(f"{p}.{j}", N)
        for j in (4, 10, 15):
            if mod == "rgb":
                spf_single(f"{p}.{j}", N)
        b.conv(f"{p}.5", k * N, N, 5)
        b.attention(f"{p}.9", N)
        b.conv(f"{p}.11", k * N, N, 5)
        b.conv(f"{p}.16", k * N, M, 5)
        b.attention(f"{p}.17", M)
    # synthesis.py:186-242
    for mod, cout, k in (("rgb", 3, 1), ("depth", 1, 2)):
        p = f"g_s.{mod}_synthesis_transform"
        b.attention(f"{p}.0", M)
        b.deconv(f"{p}.1", M, N, 5)
        for j in (2, 8, 13):
            if mod == "rgb":
                spf_single(f"{p}.{j}", N)
        for j in (3, 9, 14):
            b.bottleneck(f"{p}.{j}", k * N, N)
        for j in (4, 5, 10, 11, 15, 16):
            b.bottleneck(f"{p}.{j}", N)
        b.deconv(f"{p}.6", N, N, 5)
        b.attention(f"{p}.7", N)
        b.deconv(f"{p}.12", N, N, 5)
        b.deconv(f"{p}.17", N, cout, 5)
    for mod in ("rgb", "depth"):  # analysis.py:231-237 (HyperAnalysisEXcross, as in ELIC_united)
        p = f"h_a.{mod}_reduction"
        b.conv(f"{p}.0", M, N, 3)
        b.conv(f"{p}.2", N, N, 5)
        b.conv(f"{p}.4", N, N, 5)
    # synthesis.py:325-380: RGB hyper synthesis on its own, depth on (depth, rgb)
    for m, mult in (("r", 1), ("d", 2)):
        for idx, (cin, cout, k) in enumerate(((N, M, 5), (M, M * 3 // 2, 5), (M * 3 // 2, 2 * M, 3)), 1):
            p = f"h_s.{m}_h_s{idx}"
            b.se(f"{p}.se", mult * cin)
            b.deconv(f"{p}.deconv", mult * cin, cout, k)
    for fam in ("rgb_local_context", "rgb_local_context_anchor_with_nonanchor", "depth_local_context"):
        for i, c in enumerate(slice_ch):
            b.conv(f"{fam}.{i}", c, 2 * c, 5)
    for fam in ("rgb_channel_context", "depth_channel_context"):
        for i in range(1, len(slice_ch)):
            b.channel_context(f"{fam}.{i}", sum(slice_ch[:i]), 2 * slice_ch[i])
    for fam, key in (("rgb_entropy_parameters_anchor", "rgb_anchor"), ("depth_entropy_parameters_anchor", "depth_anchor"),
                     ("rgb_entropy_parameters_nonanchor", "rgb_nonanchor"),
                     ("depth_entropy_parameters_nonanchor", "depth_nonanchor")):
        for i, c in enumerate(slice_ch):
            b.entropy_params(f"{fam}.{i}", r2d_entropy_param_in_dims(M, slice_ch, i)[key], 2 * c)
    b.entropy_bottleneck("rgb_entropy_bottleneck", N)
    b.entropy_bottleneck("depth_entropy_bottleneck", N)
    b.gaussian_conditional("rgb_gaussian_conditional")
    b.gaussian_conditional("depth_gaussian_conditional")
    return b.entries


STF_DEPTHS = (2, 2, 6, 2)
STF_HEADS = (3, 6, 12, 24)
STF_EMBED = 48


def stf_config() -> Config:
    """models/stf_united.py:638-640: the Swin variant overrides N, M and the slice widths."""
    cfg = model_config()
    cfg["N"], cfg["M"], cfg["slice_ch"] = 192, 384, [24, 24, 48, 96, 192]
    return cfg


def stf_united_entries(config=None) -> "OrderedDict[str, Entry]":
    """Every state_dict entry of STF_united (reference: models/stf_united.py:403-678; 1440 tensors): the Swin analysis /
    synthesis transforms plus ELIC_united's hyper and entropy nets at N=192, M=384, slices [24,24,48,96,192]."""
    cfg = stf_config()
    base = elic_united_entries(cfg)
    b = _Builder()
    E, depths, heads = STF_EMBED, STF_DEPTHS, STF_HEADS
    # analysis: stf_united.py:403-502
    for mod, cin in (("rgb", 3), ("depth", 1)):
        b.conv(f"g_a.{mod}_patch_embed.proj", cin, E, 2)
        b.layernorm(f"g_a.{mod}_patch_embed.norm", E)
    for mod in ("rgb", "depth"):
        dim, li = E, 0
        for i in range(4):
            p = f"g_a.{mod}_ana_layers.{li}"
            for k in range(depths[i]):
                b.swin_block(f"{p}.blocks.{k}", dim, heads[i])
            if i < 3:
                b.linear(f"{p}.downsample.reduction", 4 * dim, 2 * dim, bias=False)
                b.layernorm(f"{p}.downsample.norm", 4 * dim)
            dim *= 2
            li += 1
            if i < 3:
                if mod == "rgb":
                    b.bi_spf(f"g_a.rgb_ana_layers.{li}", dim)
                li += 1
    # synthesis: stf_united.py:505-602
    for mod, cout in (("rgb", 3), ("depth", 1)):
        dim, li = 8 * E, 0
        for i in range(4):
            p = f"g_s.{mod}_syn_layers.{li}"
            for k in range(depths[3 - i]):
                b.swin_block(f"{p}.blocks.{k}", dim, heads[3 - i])
            if i < 3:
                b.linear(f"{p}.downsample.reduction", dim, 2 * dim, bias=False)
                b.layernorm(f"{p}.downsample.norm", dim)
            dim //= 2
            li += 1
            if i < 3:
                if mod == "rgb":
                    b.bi_spf(f"g_s.rgb_syn_layers.{li}", dim)
                li += 1
        b.conv(f"g_s.{mod}_end_conv.0", E, 4 * E, 5)
        b.conv(f"g_s.{mod}_end_conv.2", E, cout, 3)
    out = OrderedDict(b.entries)
    for k, v in base.items():
        if not k.startswith(("g_a.", "g_s.")):
            out[k] = v
    return out


def count_parameters(entries) -> int:
    n = 0
    for e in entries.values():
        if e.is_param:
            k = 1
            for s in e.shape:
                k *= s
            n += k
    return n
