"""Deterministic synthetic weights and images (no checkpoint ships with the reference).

Values come from numpy's Philox counter generator through integer arithmetic only, so the same
(name, seed) gives the same float32 tensor on any machine / numpy / torch version:
    raw 64-bit word -> four 16-bit fields -> their integer sum (Irwin-Hall, ~Gaussian) -> scale -> float32.

The "stress" recipe multiplies a few tensors so that the latents leave the dead zone
and the rANS coder sees a non-trivial symbol distribution (dozens of scale bins, escape symbols).
"""
import hashlib
import math
from collections import OrderedDict

import numpy as np

from .arch import (Entry, elic_entries, elic_united_entries, elic_united_r2d_entries, model_config, stf_config,
                   stf_united_entries)

_IH_STD = math.sqrt(4.0 * (65536.0**2 - 1.0) / 12.0)  # std of the sum of four uniform 16-bit ints
_IH_MEAN = 2.0 * 65535.0


def _key(name: str, seed: int) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:8], "little")


def _raw(name: str, seed: int, n: int) -> np.ndarray:
    return np.random.Philox(key=_key(name, seed)).random_raw(n)


def normal_like(name: str, seed: int, shape, std: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    w = _raw(name, seed, n)
    s = (w & 0xFFFF) + ((w >> 16) & 0xFFFF) + ((w >> 32) & 0xFFFF) + (w >> 48)
    v = (s.astype(np.float64) - _IH_MEAN) * (std / _IH_STD)
    return v.astype(np.float32).reshape(shape)


def uniform_like(name: str, seed: int, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    w = _raw(name, seed, n)
    u = (w >> 11).astype(np.float64) * (1.0 / 9007199254740992.0)  # 53-bit mantissa, [0,1)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def _eb_matrix_init(shape, index: int, filters=(3, 3, 3, 3), init_scale=10.0) -> np.ndarray:
    # entropy_models.py:291-299: log(expm1(1 / scale / filters[i+1]))
    f = (1,) + tuple(filters) + (1,)
    scale = init_scale ** (1.0 / (len(filters) + 1))
    init = math.log(math.expm1(1.0 / scale / f[index + 1]))
    return np.full(shape, init, dtype=np.float32)


def make_tensor(name: str, e: Entry, seed: int) -> np.ndarray:
    if e.kind in ("conv_w", "deconv_w", "bias"):
        # torch's default Conv2d/ConvTranspose2d init, U(-1/sqrt(fan_in), 1/sqrt(fan_in)); the reference's
        # kaiming pass (priors.py:63-68) runs before its conv layers exist, so this is what it starts from
        b = 1.0 / math.sqrt(e.fan_in)
        return uniform_like(name, seed, e.shape, -b, b)
    if e.kind == "linear_w":
        b = 1.0 / math.sqrt(e.fan_in)
        return uniform_like(name, seed, e.shape, -b, b)
    if e.kind == "ln_w":  # nn.LayerNorm starts at (1, 0); perturbed so that the affine part is exercised
        return (1.0 + uniform_like(name, seed, e.shape, -0.1, 0.1)).astype(np.float32)
    if e.kind == "ln_b":
        return uniform_like(name, seed, e.shape, -0.1, 0.1)
    if e.kind == "rpb_table":  # trunc_normal_(std=0.02) in the reference; wider here so the bias matters
        return normal_like(name, seed, e.shape, 0.3)
    if e.kind == "eb_matrix":
        idx = int(name[-1])
        return _eb_matrix_init(e.shape, idx) + normal_like(name, seed, e.shape, 0.05)
    if e.kind == "eb_bias":
        return uniform_like(name, seed, e.shape, -0.5, 0.5)
    if e.kind == "eb_factor":
        return normal_like(name, seed, e.shape, 0.1)
    if e.kind == "eb_quantiles":
        c = e.shape[0]
        med = uniform_like(name + "#med", seed, (c,), -0.4, 0.4)
        lo = uniform_like(name + "#lo", seed, (c,), 7.0, 12.0)
        hi = uniform_like(name + "#hi", seed, (c,), 7.0, 12.0)
        q = np.stack([med - lo, med, med + hi], axis=1).astype(np.float32)
        return q.reshape(e.shape)
    if e.kind == "buffer":
        if e.dtype == "int32":
            return np.zeros(e.shape, dtype=np.int32)
        if e.dtype == "int64":  # relative_position_index of a Swin window (stf_united.py:62-72)
            ws = int(round(math.sqrt(e.shape[0])))
            c = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")).reshape(2, -1)
            rel = (c[:, :, None] - c[:, None, :]).transpose(1, 2, 0) + (ws - 1)
            return (rel[:, :, 0] * (2 * ws - 1) + rel[:, :, 1]).astype(np.int64)
        if name.endswith(".target"):
            t = math.log(2.0 / 1e-9 - 1.0)  # entropy_models.py:309-310
            return np.array([-t, 0.0, t], dtype=np.float32)
        if name.endswith("likelihood_lower_bound.bound"):
            return np.array([1e-9], dtype=np.float32)
        if name.endswith("lower_bound_scale.bound") or name.endswith("scale_bound"):
            return np.array([0.11], dtype=np.float32)
        return np.zeros(e.shape, dtype=np.float32)
    raise ValueError(f"unknown entry kind {e.kind} for {name}")


def synthetic_state_dict(seed: int = 0, config=None, stress: bool = True, as_torch: bool = True,
                         model: str = "ELIC_united", channel: int = 3, recipe: str = None):
    """Full state_dict (parameters + buffers) of ELIC_united (default) or the single-modal ELIC with deterministic
    synthetic values.  `recipe`: "stress" (= stress=True, the default: ~22 bpp, wide CDF rows, 17 % escapes -- the worst
    case for the entropy coder), "trained_like" (ELIC_united only: latents mostly inside the dead zone, scales near the
    bottom of the scale table, ~1 bpp per modality like a trained q=2_2 model -- the coder's realistic operating point),
    "high_rate" (ELIC_united only: latents of tens to hundreds, predicted scales of 10 ... 100 -- scale-table rows of 300 ...
    3000 entries, what a high-quality checkpoint makes the decoder search) or "plain" (default initialisation, everything
    quantises to zero)."""
    if recipe is not None:
        if recipe not in ("stress", "trained_like", "high_rate", "plain"):
            raise ValueError(f"unknown recipe {recipe}")
        stress = recipe == "stress"
    if model == "STF_united":
        cfg = stf_config()
        entries = stf_united_entries()
    else:
        cfg = model_config() if config is None else config
        entries = {"ELIC_united": elic_united_entries, "ELIC_united_R2D": elic_united_r2d_entries}.get(model)
        entries = entries(cfg) if entries else elic_entries(cfg, channel)
    sd = OrderedDict()
    for name, e in entries.items():
        sd[name] = make_tensor(name, e, seed)
    if stress and model in ("ELIC_united", "ELIC_united_R2D"):
        _apply_stress(sd, cfg)
    elif stress and model == "STF_united":
        _apply_stress(sd, cfg, transforms=False)
        for mod in ("rgb", "depth"):  # latents of a few units: the last patch-merging projection feeds the final stage
            sd[f"g_a.{mod}_ana_layers.4.downsample.reduction.weight"] *= np.float32(STF_Y_GAIN)
    elif stress:
        _apply_stress_single(sd, cfg)
    if recipe == "trained_like":
        if model != "ELIC_united":
            raise ValueError("the trained_like recipe is defined for ELIC_united")
        _apply_trained_like(sd, cfg)
    if recipe == "high_rate":
        if model != "ELIC_united":
            raise ValueError("the high_rate recipe is defined for ELIC_united")
        _apply_high_rate(sd, cfg)
    if as_torch:
        import torch

        return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd.items())
    return sd


STF_Y_GAIN = 12.0


def _apply_stress(sd, cfg, transforms=True):
    slice_ch = list(cfg["slice_ch"])
    for mod in ("rgb", "depth"):
        if transforms:
            sd[f"g_a.{mod}_analysis_transform.16.weight"] *= np.float32(48.0)
    for mod in ("rgb", "depth"):
        sd[f"h_a.{mod}_reduction.4.weight"] *= np.float32(24.0)  # |z| of a few units: exercises the z coder
    for m in ("r", "d"):
        sd[f"h_s.{m}_h_s3.deconv.weight"] *= np.float32(40.0)
    for fam in ("rgb_entropy_parameters_anchor", "depth_entropy_parameters_anchor",
                "rgb_entropy_parameters_nonanchor", "depth_entropy_parameters_nonanchor"):
        for i, c in enumerate(slice_ch):
            sd[f"{fam}.{i}.fusion.4.weight"] *= np.float32(6.0)
            sd[f"{fam}.{i}.fusion.4.bias"][:c] = np.float32(1.0)  # scale half


def _apply_trained_like(sd, cfg):
    """Rates of a trained low-rate model instead of the stress recipe's: |y| of a few tenths (most symbols round to zero,
    some to +-1/+-2), predicted scales around 0.2-0.4 (scale-table rows of 7-9 entries), small |z|."""
    slice_ch = list(cfg["slice_ch"])
    for mod in ("rgb", "depth"):
        sd[f"g_a.{mod}_analysis_transform.16.weight"] *= np.float32(6.0)
        sd[f"h_a.{mod}_reduction.4.weight"] *= np.float32(6.0)
    for m in ("r", "d"):
        sd[f"h_s.{m}_h_s3.deconv.weight"] *= np.float32(8.0)
    for fam in ("rgb_entropy_parameters_anchor", "depth_entropy_parameters_anchor",
                "rgb_entropy_parameters_nonanchor", "depth_entropy_parameters_nonanchor"):
        for i, c in enumerate(slice_ch):
            sd[f"{fam}.{i}.fusion.4.weight"] *= np.float32(2.0)
            sd[f"{fam}.{i}.fusion.4.bias"][:c] = np.float32(0.25)  # scale half


HIGH_RATE_GAINS = (240.0, 24.0, 40.0, 12.0, 40.0)  # y, z, h_s, scale-head weight, scale-head bias


def _apply_high_rate(sd, cfg):
    """Wide scale-table rows: |y| of tens to hundreds and predicted scales of 10 ... 100 (sigma-index 37 ... 56 of 64, CDF rows
    of 300 ... 3000 entries) -- the rows a high-quality checkpoint codes on, which neither other recipe reaches
    (sigma-index <= 33).  Applied to the plain initialisation."""
    gy, gz, gh, gw, gb = (np.float32(v) for v in HIGH_RATE_GAINS)
    slice_ch = list(cfg["slice_ch"])
    for mod in ("rgb", "depth"):
        sd[f"g_a.{mod}_analysis_transform.16.weight"] *= gy
        sd[f"h_a.{mod}_reduction.4.weight"] *= gz
    for m in ("r", "d"):
        sd[f"h_s.{m}_h_s3.deconv.weight"] *= gh
    for fam in ("rgb_entropy_parameters_anchor", "depth_entropy_parameters_anchor",
                "rgb_entropy_parameters_nonanchor", "depth_entropy_parameters_nonanchor"):
        for i, c in enumerate(slice_ch):
            sd[f"{fam}.{i}.fusion.4.weight"] *= gw
            sd[f"{fam}.{i}.fusion.4.bias"][:c] = gb  # scale half


def _apply_stress_single(sd, cfg):
    # same idea as _apply_stress for the single-modal ELIC (its entropy-parameter nets end in a 1x1 convolution)
    slice_ch = list(cfg["slice_ch"])
    sd["g_a.analysis_transform.13.weight"] *= np.float32(48.0)
    sd["h_a.reduction.4.weight"] *= np.float32(24.0)
    sd["h_s.increase.4.weight"] *= np.float32(40.0)
    for fam in ("entropy_parameters_anchor", "entropy_parameters_nonanchor"):
        for i, c in enumerate(slice_ch):
            sd[f"{fam}.{i}.fusion.4.weight"] *= np.float32(6.0)
            sd[f"{fam}.{i}.fusion.4.bias"][:c] = np.float32(1.0)  # scale half


def synthetic_pair(index: int, H: int, W: int, config_id: int = 0, smooth: bool = False):
    """One RGB-D pair in [0,1): rgb [3,H,W], depth [1,H,W] float32 (seed = 1000*config + index)."""
    seed = 1000 * config_id + index
    if not smooth:
        rgb = uniform_like("rgb", seed, (3, H, W), 0.0, 1.0)
        depth = uniform_like("depth", seed, (1, H, W), 0.0, 1.0)
        return rgb, depth
    # spatially correlated variant: bilinear-interpolated 8x8 noise (pure float64 arithmetic)
    def up(name, ch):
        g = uniform_like(name, seed, (ch, 9, 9), 0.0, 1.0).astype(np.float64)
        ys = np.linspace(0.0, 8.0, H, endpoint=False)
        xs = np.linspace(0.0, 8.0, W, endpoint=False)
        y0 = np.floor(ys).astype(int)
        x0 = np.floor(xs).astype(int)
        fy = (ys - y0)[None, :, None]
        fx = (xs - x0)[None, None, :]
        a = g[:, y0][:, :, x0]
        b = g[:, y0][:, :, x0 + 1]
        c = g[:, y0 + 1][:, :, x0]
        d = g[:, y0 + 1][:, :, x0 + 1]
        return ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy).astype(np.float32)

    return up("rgb_s", 3), up("depth_s", 1)


def synthetic_batch(B: int, H: int, W: int, config_id: int = 0, start: int = 0, smooth: bool = False):
    rs, ds = [], []
    for i in range(B):
        r, d = synthetic_pair(start + i, H, W, config_id, smooth)
        rs.append(r)
        ds.append(d)
    return np.stack(rs), np.stack(ds)


def synthetic_latents(B: int, h: int, w: int, M: int = 320, seed: int = 0):
    """Inputs of the Bi-CEE stage in isolation (BASELINE config 4, SURVEY §8d "C4"): y ~ N(0, 4^2) [B,M,h,w] and hyper
    parameters ~ N(0, 1) [B,2M,h,w] per modality.  Returns (rgb_y, rgb_hyper, depth_y, depth_hyper) as float32."""
    return (normal_like(f"c4.rgb_y.{B}x{h}x{w}", seed, (B, M, h, w), 4.0),
            normal_like(f"c4.rgb_hyper.{B}x{h}x{w}", seed, (B, 2 * M, h, w), 1.0),
            normal_like(f"c4.depth_y.{B}x{h}x{w}", seed, (B, M, h, w), 4.0),
            normal_like(f"c4.depth_hyper.{B}x{h}x{w}", seed, (B, 2 * M, h, w), 1.0))
