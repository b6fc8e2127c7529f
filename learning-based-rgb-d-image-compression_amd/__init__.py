"""MI355X-native ELIC_united encode/decode hot path.

The directory name is not a Python identifier; import it through the `rgbd_amd` alias package at the
repository root (`import rgbd_amd`), which resolves to this package.
"""
from . import arch, synth  # noqa: F401
from .arch import Config, model_config  # noqa: F401
from . import _lib, ans, distributed, entropy_models  # noqa: F401,E402
from .elic_united import ELIC_united, modelZoo  # noqa: F401,E402
from .elic import ELIC  # noqa: F401,E402

from .stf_united import STF_united, SymmetricalTransFormerUnited  # noqa: F401,E402

modelZoo["ELIC"] = ELIC
modelZoo["STF_united"] = SymmetricalTransFormerUnited  # models/__init__.py:11-20
from .pool import CodecPool  # noqa: F401,E402
from . import datautils, ioutils, metrics, tester  # noqa: F401,E402
from .tester import TesterSingle, TesterUnited  # noqa: F401,E402
