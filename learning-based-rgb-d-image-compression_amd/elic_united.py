"""`ELIC_united` on MI355X: the reference's model API (models/elic_united.py) over the HIP engine.

Drop-in surface for testing/tester.py:55-108 and testing/tester_united.py:141-195:
    net = ELIC_united(config=model_config(), channel=4).eval()
    net.load_state_dict(checkpoint["state_dict"]); net.update(force=True); net = net.to("cuda")
    out = net.compress(rgb, depth)          -> {"r_strings": [[y], [z]*B], "d_strings": ..., "shape": (H/64, W/64)}
    rec = net.decompress(out["r_strings"], out["d_strings"], out["shape"])  -> {"x_hat": {"r","d"}, "cost_time"}

The network, the checkerboard entropy model and the rANS coder all run inside librgbd_amd.so (hand-written gfx950
kernels); this class only holds the checkpoint, builds the integer tables once (update()) and marshals pointers.
There is no CPU execution path: compress()/decompress() raise if the HIP library or a GPU is missing.
"""
import ctypes
import logging
import time
from collections import OrderedDict

import numpy as np
import torch

from . import synth
from ._lib import RgbdError, check, lib
from .arch import elic_united_entries, model_config
from .entropy_models import EntropyBottleneck, GaussianConditional, get_scale_table

_TABLE_KEYS = ("_offset", "_quantized_cdf", "_cdf_length")
_log = logging.getLogger("rgbd_amd")


class ELIC_united:
    def __init__(self, config=None, channel=4, init_seed=0, **kwargs):
        self.config = model_config() if config is None else config
        self.channel = channel
        self.N, self.M = int(self.config["N"]), int(self.config["M"])
        self.slice_ch = list(self.config["slice_ch"])
        self.slice_num = len(self.slice_ch)
        self.quant = self.config.get("quant", "ste") if hasattr(self.config, "get") else "ste"
        self.training = False
        self.per_image_streams = False  # False = reference format: one y-stream per modality for the whole batch
        self._entries = elic_united_entries(self.config)
        self._init_seed = init_seed
        self._params = None  # name -> torch CPU tensor (parameters and float buffers)
        self.rgb_gaussian_conditional = GaussianConditional(None)
        self.depth_gaussian_conditional = GaussianConditional(None)
        self._store = _LazyStore(self)
        self.rgb_entropy_bottleneck = EntropyBottleneck(self._store, "rgb_entropy_bottleneck")
        self.depth_entropy_bottleneck = EntropyBottleneck(self._store, "depth_entropy_bottleneck")
        self._h = None
        self._device = None
        self._dirty = True
        self._gen = 0        # bumped by every upload of weights / tables to the GPU
        self._parent = None  # set on shared-weight clones (clone_shared)

    # ---- torch.nn.Module-like surface ------------------------------------------------------------------
    _MODEL = "ELIC_united"

    def _materialize(self):
        if self._params is None:
            self._params = synth.synthetic_state_dict(self._init_seed, self.config, stress=False, model=self._MODEL)
        return self._params

    def _holders(self):
        """state_dict prefix -> table holder."""
        return {"rgb_gaussian_conditional": self.rgb_gaussian_conditional,
                "depth_gaussian_conditional": self.depth_gaussian_conditional,
                "rgb_entropy_bottleneck": self.rgb_entropy_bottleneck,
                "depth_entropy_bottleneck": self.depth_entropy_bottleneck}

    def _table_slots(self):
        """(engine table slot, holder): 0/1 gaussian rgb/depth, 2/3 bottleneck rgb/depth."""
        return [(0, self.rgb_gaussian_conditional), (1, self.depth_gaussian_conditional),
                (2, self.rgb_entropy_bottleneck), (3, self.depth_entropy_bottleneck)]

    def _create_engine(self):
        h = ctypes.c_void_p()
        sl = (ctypes.c_int32 * len(self.slice_ch))(*self.slice_ch)
        check(lib().rgbd_elic_create(self.N, self.M, sl, len(self.slice_ch), ctypes.byref(h)), "elic_create")
        return h

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("the MI355X path is inference-only (training is out of scope, DESIGN.md)")
        return self

    def parameters(self):
        p = self._materialize()
        for name, e in self._entries.items():
            if e.is_param:
                yield p[name]

    def count_parameters(self, only_trainable=False):
        return sum(p.numel() for p in self.parameters())

    def state_dict(self):
        p = self._materialize()
        out = OrderedDict()
        holders = self._holders()
        for name in self._entries:
            mod, _, leaf = name.rpartition(".")
            if mod in holders and leaf in _TABLE_KEYS:
                out[name] = getattr(holders[mod], leaf)
            elif mod in holders and leaf == "scale_table":
                out[name] = holders[mod].scale_table
            else:
                out[name] = p[name]
        return out

    def load_state_dict(self, state_dict, strict=False):
        """Accepts the reference's key set.  Like the reference (models/elic_united.py:613-620:
        strict first, traceback printed, then strict=False) a partial checkpoint is accepted but never silently: the
        missing / unexpected key lists are logged, a DDP "module." prefix is stripped, and a checkpoint that matches no
        parameter at all raises instead of leaving the model on its synthetic initialisation."""
        self._require_owner("load_state_dict")
        if state_dict and all(k.startswith("module.") for k in state_dict):  # saved from a DDP / DataParallel wrapper
            state_dict = OrderedDict((k[len("module."):], v) for k, v in state_dict.items())
        missing = [k for k in self._entries if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._entries]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        n_params = sum(1 for k, e in self._entries.items() if e.is_param)
        hit = sum(1 for k, e in self._entries.items() if e.is_param and k in state_dict)
        if n_params and not hit:
            raise RuntimeError(f"load_state_dict: none of the {n_params} parameter keys of {self._MODEL} is in the checkpoint "
                               f"(first checkpoint keys: {list(state_dict)[:3]})")
        if missing:
            _log.warning("%s.load_state_dict: %d missing keys keep their initial values: %s%s", self._MODEL, len(missing),
                         missing[:8], " ..." if len(missing) > 8 else "")
        if unexpected:
            _log.warning("%s.load_state_dict: %d unexpected keys ignored: %s%s", self._MODEL, len(unexpected),
                         unexpected[:8], " ..." if len(unexpected) > 8 else "")
        params = self._materialize() if missing else OrderedDict()
        holders = self._holders()
        for name, e in self._entries.items():
            if name not in state_dict:
                continue
            v = state_dict[name]
            v = v.detach().cpu() if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            mod, _, leaf = name.rpartition(".")
            if mod in holders and leaf in _TABLE_KEYS:
                setattr(holders[mod], leaf, v.to(torch.int32).clone())
                continue
            if mod in holders and leaf == "scale_table":
                holders[mod].scale_table = v.float().clone()
                continue
            if e.shape and tuple(v.shape) != tuple(e.shape) and e.kind != "buffer":
                raise RuntimeError(f"size mismatch for {name}: checkpoint {tuple(v.shape)} vs model {tuple(e.shape)}")
            params[name] = v.float().contiguous().clone() if v.is_floating_point() else v.clone()
        self._params = params
        self._dirty = True
        return None

    def update(self, scale_table=None, force=False):
        """models/elic_united.py:580-586 + compressai/models/priors.py:73-92."""
        self._require_owner("update")
        self._materialize()
        if scale_table is None:
            scale_table = get_scale_table()
        r = self.rgb_gaussian_conditional.update_scale_table(scale_table, force=force)
        d = self.depth_gaussian_conditional.update_scale_table(scale_table, force=force)
        eb = False
        for m in (self.rgb_entropy_bottleneck, self.depth_entropy_bottleneck):
            eb |= bool(m.update(force=force))
        self._dirty = True
        return (r & d) | eb

    def to(self, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RgbdError("ELIC_united (rgbd_amd) runs on the GPU only: use .to('cuda'); there is no CPU path")
        if not torch.cuda.is_available():
            raise RgbdError("no HIP device visible to torch")
        self._device = torch.device("cuda", torch.cuda.current_device() if dev.index is None else dev.index)
        self._upload()
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else f"cuda:{int(device)}")

    # ---- engine ------------------------------------------------------------------------------------
    def _upload(self):
        L = lib()
        torch.cuda.set_device(self._device)
        if self._h is None:
            self._h = self._create_engine()
        p = self._materialize()
        for name, e in self._entries.items():
            if not e.is_param:
                continue
            t = p[name].detach().float().contiguous()
            a = t.numpy()
            if e.kind == "linear_w" and ".fc." not in name:  # nn.Linear of the Swin blocks -> 1x1 convolution
                a = a.reshape(a.shape[0], a.shape[1], 1, 1)
            shape = (ctypes.c_int64 * a.ndim)(*a.shape)
            check(L.rgbd_elic_set_tensor(self._h, name.encode(), a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), shape,
                                         a.ndim), f"set_tensor({name})")
        for which, holder in self._table_slots():
            cdf, sizes, offs = holder.numpy_tables()  # raises "Uninitialized CDFs. Run update() first"
            i32p = ctypes.POINTER(ctypes.c_int32)
            check(L.rgbd_elic_set_tables(self._h, which, cdf.ctypes.data_as(i32p), int(cdf.shape[1]),
                                         sizes.ctypes.data_as(i32p), offs.ctypes.data_as(i32p), int(cdf.shape[0])),
                  f"set_tables({which})")
        st = self._table_slots()[0][1].scale_table.float().contiguous().numpy()
        check(L.rgbd_elic_set_scale_table(self._h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(st.shape[0])),
              "set_scale_table")
        from . import refarith

        refarith.push(L, self._h, check)  # the reference's accumulation blocks per layer shape
        check(L.rgbd_elic_finalize(self._h), "finalize")
        self._dirty = False
        self._gen += 1

    def _require_owner(self, what):
        if self._parent is not None:
            raise RgbdError(f"{what}() on a shared-weight clone: call it on the parent engine (clones follow it)")

    def _ready(self):
        if self._h is None or self._device is None:
            raise RgbdError("call .to('cuda') before compress()/decompress()")
        if self._parent is not None:
            # shared-weight clone: follow the parent.  If the parent has re-uploaded weights or tables since this clone
            # was made (update() / load_state_dict()), take a fresh clone of it -- the old device buffers stay valid
            # until then (they are reference-counted in the library), they are merely stale.
            par = self._parent
            par._ready()
            if par._gen != self._gen:
                L = lib()
                h = ctypes.c_void_p()
                check(L.rgbd_elic_clone_shared(par._h, ctypes.byref(h)), "clone_shared")
                L.rgbd_elic_destroy(self._h)
                self._h = h
                self._gen = par._gen
                self._params = par._params
                if getattr(self, "_tile_mode", None):
                    check(L.rgbd_elic_set_tile_mode(self._h, {"latency": 0, "throughput": 1}[self._tile_mode]), "tile_mode")
            torch.cuda.set_device(self._device)
            return
        if self._dirty:
            self._upload()
        torch.cuda.set_device(self._device)

    @staticmethod
    def _stream_ptr():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def compress(self, rgb, depth):
        self._ready()
        if rgb.dim() != 4 or depth.dim() != 4 or rgb.size(1) != 3 or depth.size(1) != 1:
            raise ValueError("expected rgb [B,3,H,W] and depth [B,1,H,W]")
        B, _, H, W = rgb.shape
        if depth.shape[0] != B or depth.shape[-2:] != rgb.shape[-2:]:
            raise ValueError("rgb and depth must have the same batch and spatial size")
        if H % 64 or W % 64:
            raise ValueError("H and W must be multiples of 64 (pad first: dataset/utils.py:58-67)")
        rgb = rgb.to(self._device, torch.float32).contiguous()
        depth = depth.to(self._device, torch.float32).contiguous()
        L = lib()
        check(L.rgbd_elic_compress(self._h, ctypes.c_void_p(rgb.data_ptr()), ctypes.c_void_p(depth.data_ptr()), B, H, W,
                                   1 if self.per_image_streams else 0, self._stream_ptr()), "compress")
        out = [[self._fetch_streams(mod, kind) for kind in (0, 1)] for mod in (0, 1)]
        return {"r_strings": out[0], "d_strings": out[1], "shape": torch.Size((H // 64, W // 64))}

    def _fetch_streams(self, mod, kind):
        L = lib()
        strings = []
        for i in range(L.rgbd_elic_stream_count(self._h, mod, kind)):
            p = ctypes.POINTER(ctypes.c_uint8)()
            ln = ctypes.c_int64(0)
            check(L.rgbd_elic_stream(self._h, mod, kind, i, ctypes.byref(p), ctypes.byref(ln)), "stream")
            strings.append(ctypes.string_at(p, ln.value))
        return strings

    @staticmethod
    def _pack_strings(strings):
        bufs = [np.frombuffer(bytes(s), dtype=np.uint8) for s in strings]
        u8p = ctypes.POINTER(ctypes.c_uint8)
        ptrs = (u8p * len(bufs))(*[b.ctypes.data_as(u8p) for b in bufs])
        lens = (ctypes.c_int64 * len(bufs))(*[int(b.shape[0]) for b in bufs])
        return bufs, ptrs, lens

    # ---- the Bi-CEE stage alone (models/elic_united.py:350-401, 543-578) ------------------------------------
    def _check_latents(self, y, hyper, what):
        if y is not None and (y.dim() != 4 or y.size(1) != self.M):
            raise ValueError(f"{what} y: expected [B,{self.M},h,w]")
        if hyper.dim() != 4 or hyper.size(1) != 2 * self.M:
            raise ValueError(f"{what} hyper parameters: expected [B,{2 * self.M},h,w]")
        if y is not None and (y.shape[0] != hyper.shape[0] or y.shape[-2:] != hyper.shape[-2:]):
            raise ValueError(f"{what}: y and hyper parameters disagree in batch or spatial size")
        if hyper.shape[-1] % 2:
            raise ValueError("latent width must be even (checkerboard packing, utils/ckbd.py:51-64)")

    def compress_united(self, rgb_y, rgb_hyper_params, depth_y, depth_hyper_params):
        """Latents + hyper parameters -> (rgb_y_strings, depth_y_strings), each a list of byte strings (one for the whole
        batch like the reference, or one per image when per_image_streams is set)."""
        self._ready()
        self._check_latents(rgb_y, rgb_hyper_params, "rgb")
        self._check_latents(depth_y, depth_hyper_params, "depth")
        if rgb_y.shape != depth_y.shape:
            raise ValueError("rgb and depth latents must have the same shape")
        t = [x.to(self._device, torch.float32).contiguous() for x in (rgb_y, rgb_hyper_params, depth_y, depth_hyper_params)]
        B, _, h, w = t[0].shape
        p = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731
        check(lib().rgbd_elic_compress_united(self._h, p(t[0]), p(t[1]), p(t[2]), p(t[3]), B, h, w,
                                              1 if self.per_image_streams else 0, self._stream_ptr()), "compress_united")
        return self._fetch_streams(0, 0), self._fetch_streams(1, 0)

    def decompress_united(self, rgb_y_strings, rgb_hyper_params, depth_y_strings, depth_hyper_params):
        """(y strings, hyper parameters) per modality -> (rgb_y_hat, depth_y_hat).  A string argument may be one bytes
        object (what the reference passes: strings[0][0]) or a list of them (one per image)."""
        self._ready()
        self._check_latents(None, rgb_hyper_params, "rgb")
        self._check_latents(None, depth_hyper_params, "depth")
        y_r = [rgb_y_strings] if isinstance(rgb_y_strings, (bytes, bytearray)) else list(rgb_y_strings)
        y_d = [depth_y_strings] if isinstance(depth_y_strings, (bytes, bytearray)) else list(depth_y_strings)
        hr = rgb_hyper_params.to(self._device, torch.float32).contiguous()
        hd = depth_hyper_params.to(self._device, torch.float32).contiguous()
        B, _, h, w = hr.shape
        if hd.shape != hr.shape or len(y_r) != len(y_d) or len(y_r) not in (1, B):
            raise ValueError("Invalid strings parameters")
        out_r = torch.empty((B, self.M, h, w), dtype=torch.float32, device=self._device)
        out_d = torch.empty_like(out_r)
        k1, pyr, lyr = self._pack_strings(y_r)
        k2, pyd, lyd = self._pack_strings(y_d)
        p = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731
        check(lib().rgbd_elic_decompress_united(self._h, pyr, lyr, len(y_r), pyd, lyd, p(hr), p(hd), B, h, w, p(out_r),
                                                p(out_d), self._stream_ptr()), "decompress_united")
        torch.cuda.current_stream().synchronize()
        del k1, k2
        return out_r, out_d

    def decompress(self, rgb_strings, depth_strings, shape):
        self._ready()
        # the reference brackets decompress() with torch.cuda.synchronize() (elic_united.py:431,449); here only the
        # calling stream is waited for, so engine instances on other streams (CodecPool) are not dragged into it
        torch.cuda.current_stream().synchronize()
        t0 = time.process_time()
        y_r, z_r = list(rgb_strings[0]), list(rgb_strings[1])
        y_d, z_d = list(depth_strings[0]), list(depth_strings[1])
        B = len(z_r)
        if len(z_d) != B or len(y_r) != len(y_d) or len(y_r) not in (1, B):
            raise ValueError("Invalid strings parameters")
        zh, zw = int(shape[0]), int(shape[1])
        xr = torch.empty((B, 3, zh * 64, zw * 64), dtype=torch.float32, device=self._device)
        xd = torch.empty((B, 1, zh * 64, zw * 64), dtype=torch.float32, device=self._device)

        pack = self._pack_strings
        k1, pyr, lyr = pack(y_r)
        k2, pyd, lyd = pack(y_d)
        k3, pzr, lzr = pack(z_r)
        k4, pzd, lzd = pack(z_d)
        check(lib().rgbd_elic_decompress(self._h, pyr, lyr, len(y_r), pyd, lyd, pzr, lzr, pzd, lzd, B, zh, zw,
                                         ctypes.c_void_p(xr.data_ptr()), ctypes.c_void_p(xd.data_ptr()),
                                         self._stream_ptr()), "decompress")
        torch.cuda.current_stream().synchronize()
        del k1, k2, k3, k4
        return {"x_hat": {"r": xr, "d": xd}, "cost_time": time.process_time() - t0}

    def forward(self, rgb, depth):
        """Eval-mode forward (models/elic_united.py:234-263): x_hat without entropy coding plus the likelihoods."""
        if self.training:
            raise NotImplementedError("training-mode forward (noise / STE gradients) is out of scope")
        self._ready()
        B, _, H, W = rgb.shape
        if H % 64 or W % 64:
            raise ValueError("H and W must be multiples of 64")
        rgb = rgb.to(self._device, torch.float32).contiguous()
        depth = depth.to(self._device, torch.float32).contiguous()
        dev = self._device
        xr = torch.empty((B, 3, H, W), device=dev)
        xd = torch.empty((B, 1, H, W), device=dev)
        ly = [torch.empty((B, self.M, H // 16, W // 16), device=dev) for _ in range(2)]
        lz = [torch.empty((B, self.N, H // 64, W // 64), device=dev) for _ in range(2)]
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        check(lib().rgbd_elic_forward(self._h, p(rgb), p(depth), B, H, W, p(xr), p(xd), p(ly[0]), p(ly[1]), p(lz[0]), p(lz[1]),
                                      self._stream_ptr()), "forward")
        return {"x_hat": {"r": xr, "d": xd}, "r_likelihoods": {"y": ly[0], "z": lz[0]},
                "d_likelihoods": {"y": ly[1], "z": lz[1]}}

    __call__ = forward

    # ---- parity hooks --------------------------------------------------------------------------------
    def debug_tensor(self, name: str) -> np.ndarray:
        shp = (ctypes.c_int32 * 4)()
        check(lib().rgbd_elic_debug_tensor(self._h, name.encode(), None, 0, shp), f"debug_tensor({name})")
        out = np.empty(tuple(shp), dtype=np.float32)
        check(lib().rgbd_elic_debug_tensor(self._h, name.encode(), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                           out.size, shp), f"debug_tensor({name})")
        return out

    def debug_symbols(self, modality: int):
        n = ctypes.c_int64(0)
        check(lib().rgbd_elic_debug_symbols(self._h, modality, None, None, 0, ctypes.byref(n)), "debug_symbols")
        sym = np.empty(n.value, dtype=np.int32)
        idx = np.empty(n.value, dtype=np.int32)
        i32p = ctypes.POINTER(ctypes.c_int32)
        check(lib().rgbd_elic_debug_symbols(self._h, modality, sym.ctypes.data_as(i32p), idx.ctypes.data_as(i32p),
                                            n.value, ctypes.byref(n)), "debug_symbols")
        return sym, idx

    def scale_table_numpy(self) -> np.ndarray:
        """The Gaussian scale table (float32) the engine indexes with (entropy_models.py:561-568)."""
        return self._table_slots()[0][1].scale_table.float().contiguous().numpy()

    def eb_medians_numpy(self):
        """Per-channel medians of the factorised prior(s), in modality order (entropy_models.py:437-440)."""
        return [h.medians().numpy() for which, h in self._table_slots() if which >= 2]

    def set_debug_floats(self, on: bool):
        """Keep (y - mean, scale) per symbol of the next compress() (parity bookkeeping; see debug_floats)."""
        self._ready()
        check(lib().rgbd_elic_set_debug_floats(self._h, 1 if on else 0), "set_debug_floats")

    def debug_floats(self, modality: int):
        """(x, scale) float32 arrays in stream order: what the encoder rounded / indexed for every symbol of the last
        compress() (after set_debug_floats(True))."""
        n = ctypes.c_int64(0)
        check(lib().rgbd_elic_debug_floats(self._h, modality, None, None, 0, ctypes.byref(n)), "debug_floats")
        x = np.empty(n.value, dtype=np.float32)
        sc = np.empty(n.value, dtype=np.float32)
        f32p = ctypes.POINTER(ctypes.c_float)
        check(lib().rgbd_elic_debug_floats(self._h, modality, x.ctypes.data_as(f32p), sc.ctypes.data_as(f32p), n.value,
                                           ctypes.byref(n)), "debug_floats")
        return x, sc

    def set_forced_symbols(self, modality: int, y_sym=None, z_sym=None):
        """Teacher forcing (include/rgbd_amd.h: rgbd_elic_set_forced_symbols): later contexts of the following compress()
        calls are rebuilt from these symbols (int32, stream order); None / empty clears that stage."""
        self._ready()
        i32p = ctypes.POINTER(ctypes.c_int32)
        ys = np.ascontiguousarray(y_sym if y_sym is not None else [], dtype=np.int32)
        zs = np.ascontiguousarray(z_sym if z_sym is not None else [], dtype=np.int32)
        check(lib().rgbd_elic_set_forced_symbols(self._h, modality, ys.ctypes.data_as(i32p), ys.size, zs.ctypes.data_as(i32p),
                                                 zs.size), "set_forced_symbols")

    def clone_shared(self):
        """Another engine instance on the same GPU that borrows this one's device weights and tables (own workspace and
        stream).  The clone keeps a reference to its parent so the weights outlive it."""
        self._ready()
        other = type(self).__new__(type(self))
        other.__dict__.update({k: v for k, v in self.__dict__.items() if k != "_h"})
        h = ctypes.c_void_p()
        check(lib().rgbd_elic_clone_shared(self._h, ctypes.byref(h)), "clone_shared")
        other._h = h
        other._parent = self
        other._gen = self._gen
        other._dirty = False
        return other

    def set_tile_mode(self, mode: str):
        """Convolution tile tables: "latency" (default; winners of isolated launches) or "throughput" (winners with the
        chip shared between several engine instances -- what `CodecPool` / `test_model(workers > 1)` select).  The outputs
        are bit-identical in both modes."""
        self._ready()
        self._tile_mode = mode
        check(lib().rgbd_elic_set_tile_mode(self._h, {"latency": 0, "throughput": 1}[mode]), "set_tile_mode")

    def graph_count(self) -> int:
        """Call shapes whose launch sequence is cached as a HIP graph on this engine instance."""
        return int(lib().rgbd_elic_graph_count(self._h))

    def workspace_bytes(self) -> int:
        """HBM workspace of this engine instance (not the packed weights, which the instances of a pool share)."""
        return int(lib().rgbd_elic_workspace_bytes(self._h))

    def set_profile(self, on: bool):
        check(lib().rgbd_elic_set_profile(self._h, 1 if on else 0), "set_profile")

    def profile_read(self):
        ms, n, fl = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0)
        check(lib().rgbd_elic_profile_read(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), "profile_read")
        fx = ctypes.c_double(0)
        check(lib().rgbd_elic_profile_read_executed(self._h, ctypes.byref(fx)), "profile_read_executed")
        return {"conv_ms": ms.value, "launches": n.value, "flops": fl.value, "flops_executed": fx.value}

    def __del__(self):
        try:
            if self._h is not None:
                lib().rgbd_elic_destroy(self._h)
                self._h = None
        except Exception:
            pass


class _LazyStore:
    """Mapping view used by the EntropyBottleneck table builders (parameters may be (re)loaded later)."""

    def __init__(self, owner):
        self._o = owner

    def __getitem__(self, k):
        return self._o._materialize()[k]


modelZoo = {"ELIC_united": ELIC_united}  # models/__init__.py:11-20; rgbd_amd/__init__.py adds "ELIC"
