"""`ELIC_united_R2D` on MI355X: the reference's one-directional variant (models/elic_united_R2D.py) over the HIP engine --
RGB is coded without looking at depth, depth is conditioned on RGB.  Same API and return values as ELIC_united."""
import ctypes

from ._lib import check, lib
from .arch import elic_united_r2d_entries
from .elic_united import ELIC_united


class ELIC_united_R2D(ELIC_united):
    _MODEL = "ELIC_united_R2D"

    def __init__(self, config=None, channel=4, init_seed=0, **kwargs):
        super().__init__(config=config, channel=channel, init_seed=init_seed)
        self._entries = elic_united_r2d_entries(self.config)

    def _create_engine(self):
        h = ctypes.c_void_p()
        sl = (ctypes.c_int32 * len(self.slice_ch))(*self.slice_ch)
        check(lib().rgbd_elic_create_r2d(self.N, self.M, sl, len(self.slice_ch), ctypes.byref(h)), "elic_create_r2d")
        return h

    # forward(), compress_united() and decompress_united() are inherited, exactly as in the reference
    # (models/elic_united_R2D.py overrides only the per-slice coders of models/elic_united.py): the engine created above
    # (variant "r2d") runs the one-directional slice coder and the single-input hyper synthesis behind them.
