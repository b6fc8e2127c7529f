"""`STF_united` (SymmetricalTransFormerUnited) on MI355X: the reference's Swin-transform variant of ELIC_united
(models/stf_united.py:605-678) over the HIP engine.  Same API and return values as ELIC_united; N=192, M=384 and the
slice widths [24,24,48,96,192] are fixed by the model (stf_united.py:638-640)."""
import ctypes

from ._lib import check, lib
from .arch import stf_config, stf_united_entries
from .elic_united import ELIC_united


class SymmetricalTransFormerUnited(ELIC_united):
    _MODEL = "STF_united"

    def __init__(self, config=None, channel=4, init_seed=0, **kwargs):
        super().__init__(config=stf_config(), channel=channel, init_seed=init_seed)
        self._entries = stf_united_entries()

    def _create_engine(self):
        h = ctypes.c_void_p()
        sl = (ctypes.c_int32 * len(self.slice_ch))(*self.slice_ch)
        check(lib().rgbd_elic_create_stf(self.N, self.M, sl, len(self.slice_ch), ctypes.byref(h)), "elic_create_stf")
        return h


STF_united = SymmetricalTransFormerUnited
