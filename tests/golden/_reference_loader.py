"""Import the upstream reference (read-only at /root/reference) in THIS container to generate golden vectors.

Only `tests/golden/make_golden.py` uses this; nothing here runs on the GPU box (the reference does not
travel).  Recipe = SURVEY.md §8(c)/App. D:
  * the reference's two pybind11 extensions are compiled out-of-tree into oracle/_ref/ (oracle/Makefile)
    and injected as `compressai.ans` / `compressai._CXX`;
  * three import shims for packages the image lacks (timm.models.layers, torchvision.transforms,
    PIL is present) and a no-op torch.cuda.synchronize (models/elic_united.py:431,449 call it on CPU).
"""
import importlib
import importlib.util
import os
import sys
import types

REF = os.environ.get("RGBD_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BUILD = os.path.join(ROOT, "oracle", "_ref")


def _load_ext(modname: str, stem: str):
    for fn in os.listdir(REF_BUILD):
        if fn.startswith(stem + ".") and fn.endswith(".so"):
            spec = importlib.util.spec_from_file_location(stem, os.path.join(REF_BUILD, fn))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            sys.modules[modname] = mod
            return mod
    raise FileNotFoundError(f"{stem}*.so not found in {REF_BUILD}; run `make -C oracle ref`")


def load_reference():
    """Returns (ELIC_united class, model_config function, module dict) from the unmodified reference."""
    import torch

    if not os.path.isdir(REF):
        raise RuntimeError(f"reference not present at {REF}")
    sys.dont_write_bytecode = True
    ans = _load_ext("compressai.ans", "ans")
    cxx = _load_ext("compressai._CXX", "_CXX")

    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        timm_models = types.ModuleType("timm.models")
        timm_layers = types.ModuleType("timm.models.layers")

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

        class DropPath(torch.nn.Identity):
            def __init__(self, *a, **k):
                super().__init__()

        timm_layers.to_2tuple = to_2tuple
        timm_layers.DropPath = DropPath
        timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
        timm.models = timm_models
        timm_models.layers = timm_layers
        sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers})
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")

        class ToPILImage:  # only referenced by utils/IOutils.py:8 at import time
            def __call__(self, x):
                raise NotImplementedError

        tvt.ToPILImage = ToPILImage
        tv.transforms = tvt
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt})
    if not torch.cuda.is_available():
        torch.cuda.synchronize = lambda *a, **k: None

    for p in (REF, os.path.join(REF, "CompressAI")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from config.config import model_config  # noqa: E402
    from models.elic_united import ELIC_united  # noqa: E402

    from models.elic import ELIC  # noqa: E402

    return ELIC_united, model_config, {"ans": ans, "_CXX": cxx, "ELIC": ELIC}
