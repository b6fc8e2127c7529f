"""Entropy-model tables of the ELIC_united path (host side, one-off).

`update()` of the reference builds integer CDF tables from float pmfs computed with torch CPU kernels
(CompressAI/compressai/entropy_models/entropy_models.py: EntropyBottleneck.update :320-360, GaussianConditional.update
:511-532, _pmf_to_cdf :166-172) and hands each row to the native quantiser.  The same happens here: the float pmfs use
torch CPU ops (plumbing, so the tables come out bit-identical to the reference's on the same machine) and each row goes
through the C ABI's `rgbd_pmf_to_quantized_cdf`.  The tables are then uploaded to the GPU coder; nothing here runs per
image.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ans


def get_scale_table(min=0.11, max=256, levels=64):  # noqa: A002  (utils/moduleFunc.py:11-12)
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


def _quantize_rows(pmf, tail_mass, pmf_length, max_length, precision=16):
    cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
    for i in range(pmf.shape[0]):
        row = torch.cat((pmf[i, : int(pmf_length[i])], tail_mass[i]), dim=0)
        q = ans.pmf_to_quantized_cdf(row.tolist(), precision)
        cdf[i, : len(q)] = torch.tensor(q, dtype=torch.int64).to(torch.int32)
    return cdf


class _TableHolder:
    def __init__(self):
        self._offset = torch.IntTensor()
        self._quantized_cdf = torch.IntTensor()
        self._cdf_length = torch.IntTensor()

    @property
    def offset(self):
        return self._offset

    @property
    def quantized_cdf(self):
        return self._quantized_cdf

    @property
    def cdf_length(self):
        return self._cdf_length

    def check(self):
        # same failure modes as EntropyModel._check_* (entropy_models.py:174-193)
        if self._quantized_cdf.numel() == 0:
            raise ValueError("Uninitialized CDFs. Run update() first")
        if self._quantized_cdf.dim() != 2:
            raise ValueError(f"Invalid CDF size {self._quantized_cdf.size()}")
        if self._offset.numel() == 0:
            raise ValueError("Uninitialized offsets. Run update() first")
        if self._cdf_length.numel() == 0:
            raise ValueError("Uninitialized CDF lengths. Run update() first")

    def numpy_tables(self):
        self.check()
        return (np.ascontiguousarray(self._quantized_cdf.numpy().astype(np.int32)),
                np.ascontiguousarray(self._cdf_length.reshape(-1).numpy().astype(np.int32)),
                np.ascontiguousarray(self._offset.reshape(-1).numpy().astype(np.int32)))


class GaussianConditional(_TableHolder):
    """Table side of compressai's GaussianConditional (scale_bound 0.11, tail_mass 1e-9)."""

    def __init__(self, scale_table=None, scale_bound=0.11, tail_mass=1e-9):
        super().__init__()
        self.scale_table = torch.Tensor() if scale_table is None else torch.as_tensor(scale_table, dtype=torch.float32)
        self.scale_bound = float(scale_bound)
        self.tail_mass = float(tail_mass)

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        self.scale_table = torch.as_tensor(scale_table, dtype=torch.float32).clone()
        self.update()
        return True

    def update(self):
        import scipy.stats

        mult = -scipy.stats.norm.ppf(self.tail_mass / 2)
        center = torch.ceil(self.scale_table * mult).int()
        length = 2 * center + 1
        max_length = int(torch.max(length).item())
        dist = torch.abs(torch.arange(max_length).int() - center[:, None]).float()
        sigma = self.scale_table.unsqueeze(1).float()
        c = float(-(2 ** -0.5))
        upper = 0.5 * torch.erfc(c * ((0.5 - dist) / sigma))
        lower = 0.5 * torch.erfc(c * ((-0.5 - dist) / sigma))
        self._quantized_cdf = _quantize_rows(upper - lower, 2 * lower[:, :1], length, max_length)
        self._offset = -center
        self._cdf_length = length + 2


class EntropyBottleneck(_TableHolder):
    """Table side of compressai's EntropyBottleneck(channels, filters=(3,3,3,3)); parameters live in the owner's store."""

    def __init__(self, params, prefix, filters=(3, 3, 3, 3)):
        super().__init__()
        self._p = params
        self._prefix = prefix
        self._nf = len(filters)

    def _get(self, name):
        return self._p[f"{self._prefix}.{name}"].detach().float().cpu()

    def medians(self):
        return self._get("quantiles")[:, 0, 1].contiguous()

    def _logits(self, v):
        out = v
        for i in range(self._nf + 1):
            out = torch.matmul(F.softplus(self._get(f"_matrix{i}")), out)
            out = out + self._get(f"_bias{i}")
            if i < self._nf:
                out = out + torch.tanh(self._get(f"_factor{i}")) * torch.tanh(out)
        return out

    def update(self, force=False):
        q = self._get("quantiles")
        med = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(med - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - med).int(), min=0)
        self._offset = -minima
        start = med - minima
        length = maxima + minima + 1
        max_length = int(length.max())
        samples = torch.arange(max_length)[None, :] + start[:, None, None]
        lower = self._logits(samples - 0.5)
        upper = self._logits(samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        self._quantized_cdf = _quantize_rows(pmf, tail, length, max_length)
        self._cdf_length = length + 2
        return True
