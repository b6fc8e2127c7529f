"""Single-modal `ELIC` on MI355X: the reference's model API (models/elic.py) over the HIP engine.

    net = ELIC(config=model_config(), channel=3).eval()
    net.load_state_dict(checkpoint["state_dict"]); net.update(force=True); net = net.to("cuda")
    out = net.compress(x)                              -> {"strings": [[y], [z]*B], "shape": (H/64, W/64)}
    rec = net.decompress(out["strings"], out["shape"]) -> {"x_hat": [B,C,H,W] (not clamped, as in the reference), "cost_time"}

Same engine, kernels and rules as ELIC_united (no CPU path); only the layer graph differs (no cross-modal fusion, one
entropy bottleneck, EntropyParameters = three 1x1 convolutions).
"""
import ctypes
import time

import torch

from ._lib import check, lib
from .arch import elic_entries, model_config
from .elic_united import ELIC_united, _LazyStore
from .entropy_models import EntropyBottleneck, GaussianConditional, get_scale_table


class ELIC(ELIC_united):
    _MODEL = "ELIC"

    def __init__(self, config=None, channel=3, return_mid=False, init_seed=0, **kwargs):
        if return_mid:
            raise NotImplementedError("return_mid (intermediate up-sampling outputs) is not part of the inference path")
        self.config = model_config() if config is None else config
        self.channel = channel
        self.N, self.M = int(self.config["N"]), int(self.config["M"])
        self.slice_ch = list(self.config["slice_ch"])
        self.slice_num = len(self.slice_ch)
        self.quant = self.config.get("quant", "ste") if hasattr(self.config, "get") else "ste"
        self.training = False
        self.per_image_streams = False
        self._entries = elic_entries(self.config, channel)
        self._init_seed = init_seed
        self._params = None
        self.gaussian_conditional = GaussianConditional(None)
        self._store = _LazyStore(self)
        self.entropy_bottleneck = EntropyBottleneck(self._store, "entropy_bottleneck")
        self._h = None
        self._device = None
        self._dirty = True
        self._gen = 0
        self._parent = None

    def _materialize(self):
        if self._params is None:
            from . import synth

            self._params = synth.synthetic_state_dict(self._init_seed, self.config, stress=False, model="ELIC",
                                                      channel=self.channel)
        return self._params

    def _holders(self):
        return {"gaussian_conditional": self.gaussian_conditional, "entropy_bottleneck": self.entropy_bottleneck}

    def _table_slots(self):
        return [(0, self.gaussian_conditional), (2, self.entropy_bottleneck)]

    def _create_engine(self):
        h = ctypes.c_void_p()
        sl = (ctypes.c_int32 * len(self.slice_ch))(*self.slice_ch)
        check(lib().rgbd_elic_create_single(self.N, self.M, sl, len(self.slice_ch), int(self.channel), ctypes.byref(h)),
              "elic_create_single")
        return h

    def update(self, scale_table=None, force=False):  # models/elic.py:327-332
        self._materialize()
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
        updated |= bool(self.entropy_bottleneck.update(force=force))
        self._dirty = True
        return updated

    def compress(self, x):  # models/elic.py:161-253
        self._ready()
        if x.dim() != 4 or x.size(1) != self.channel:
            raise ValueError(f"expected x [B,{self.channel},H,W]")
        B, _, H, W = x.shape
        if H % 64 or W % 64:
            raise ValueError("H and W must be multiples of 64 (pad first: dataset/utils.py:58-67)")
        x = x.to(self._device, torch.float32).contiguous()
        check(lib().rgbd_elic_compress_single(self._h, ctypes.c_void_p(x.data_ptr()), B, H, W,
                                              1 if self.per_image_streams else 0, self._stream_ptr()), "compress")
        return {"strings": [self._fetch_streams(0, 0), self._fetch_streams(0, 1)], "shape": torch.Size((H // 64, W // 64))}

    def decompress(self, strings, shape):  # models/elic.py:255-325
        self._ready()
        torch.cuda.current_stream().synchronize()
        t0 = time.process_time()
        ys, zs = list(strings[0]), list(strings[1])
        B = len(zs)
        if len(ys) not in (1, B):
            raise ValueError("Invalid strings parameters")
        zh, zw = int(shape[0]), int(shape[1])
        out = torch.empty((B, self.channel, zh * 64, zw * 64), dtype=torch.float32, device=self._device)
        k1, py, ly = self._pack_strings(ys)
        k2, pz, lz = self._pack_strings(zs)
        check(lib().rgbd_elic_decompress_single(self._h, py, ly, len(ys), pz, lz, B, zh, zw, ctypes.c_void_p(out.data_ptr()),
                                                self._stream_ptr()), "decompress")
        torch.cuda.current_stream().synchronize()
        del k1, k2
        return {"x_hat": out, "cost_time": time.process_time() - t0}

    def forward(self, x):  # models/elic.py:60-161 (eval mode, quant = "ste")
        """Eval-mode forward(): {"x_hat", "likelihoods": {"y_likelihoods", "z_likelihoods"}} like the reference (x_hat is not
        clamped; y_hat = round(y - mean) + mean slice by slice, Gaussian / factorised-prior likelihoods)."""
        self._ready()
        if self.training:
            raise RuntimeError("forward() is built for eval mode (inference path); call .eval() first")
        if self.quant != "ste":  # models/elic.py:84-99: any other setting rounds without the mean (and adds noise in training)
            raise NotImplementedError(f"forward() implements config quant = 'ste' (the reference's model_config); got {self.quant!r}")
        if x.dim() != 4 or x.size(1) != self.channel:
            raise ValueError(f"expected x [B,{self.channel},H,W]")
        B, _, H, W = x.shape
        if H % 64 or W % 64:
            raise ValueError("H and W must be multiples of 64 (pad first: dataset/utils.py:58-67)")
        x = x.to(self._device, torch.float32).contiguous()
        xh = torch.empty((B, self.channel, H, W), dtype=torch.float32, device=self._device)
        ly = torch.empty((B, self.M, H // 16, W // 16), dtype=torch.float32, device=self._device)
        lz = torch.empty((B, self.N, H // 64, W // 64), dtype=torch.float32, device=self._device)
        check(lib().rgbd_elic_forward_single(self._h, ctypes.c_void_p(x.data_ptr()), B, H, W, ctypes.c_void_p(xh.data_ptr()),
                                             ctypes.c_void_p(ly.data_ptr()), ctypes.c_void_p(lz.data_ptr()),
                                             self._stream_ptr()), "forward")
        return {"x_hat": xh, "likelihoods": {"y_likelihoods": ly, "z_likelihoods": lz}}

    __call__ = forward

    def compress_united(self, *a, **k):
        raise NotImplementedError("ELIC is single-modal")

    decompress_united = compress_united
