"""MI355X-native ELIC_united encode/decode hot path.

The directory name is not a Python identifier; import it through the `rgbd_amd` alias package at the
repository root (`import rgbd_amd`), which resolves to this package.
"""
from . import arch, synth  # noqa: F401
from .arch import Config, model_config  # noqa: F401
from . import _lib, ans, distributed, entropy_models  # noqa: F401,E402
from ._lib import RgbdError  # noqa: F401,E402
from .elic_united import ELIC_united, modelZoo  # noqa: F401,E402
from .elic import ELIC  # noqa: F401,E402

from .stf_united import STF_united, SymmetricalTransFormerUnited  # noqa: F401,E402

from .elic_united_r2d import ELIC_united_R2D  # noqa: F401,E402

# models/__init__.py:11-20 ("the complex ones first": the harness matches model names by substring)
modelZoo.clear()
modelZoo.update({"ELIC_united_R2D": ELIC_united_R2D, "ELIC_united": ELIC_united})
modelZoo["ELIC"] = ELIC
modelZoo["STF_united"] = SymmetricalTransFormerUnited  # models/__init__.py:11-20
from .pool import CodecPool  # noqa: F401,E402
from . import datautils, ioutils, metrics, tester  # noqa: F401,E402
from .tester import TesterSingle, TesterUnited  # noqa: F401,E402
