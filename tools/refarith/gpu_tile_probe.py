"""GPU box: which forced tiles reproduce the oracle bits for a blocked 1x1 / 3x3 layer (debugging aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from oracle import cpu_arith as ca
from rgbd_amd._lib import lib
import test_gpu_refarith as T
dev = torch.device("cuda:0")
x, wt, b = T._data(199, 1, 192, 32, 32, 96, 1)
want = ca.conv2d(x, wt, b, 1, 0, blocks=[112, 80], bias_mode=2)
for wm, mts, nts in ((2, (1, 2, 3, 4, 5), (1, 2, 4, 8)), (1, (1, 2, 3), (1, 2, 4))):
    for mt in mts:
        for nt in nts:
            for kc, dm in ((16, 0), (16, 1), (64, 0), (16, 4), (16, 5)):
                lib().rgbd_debug_force_tile(f"{wm},{mt},{nt},{kc},{dm}".encode())
                try:
                    got = T.gpu_conv_ref(x, wt, b, 1, 0, blocks=[112, 80], bias_mode=2, dev=dev)
                    bad = got != want
                    msg = "ok" if not bad.any() else f"BAD {int(bad.sum())} couts {sorted(set(np.nonzero(bad)[1].tolist()))[:12]}"
                except Exception as e:
                    msg = "err " + str(e)[-40:]
                print((wm, mt, nt, kc, dm), msg)
lib().rgbd_debug_force_tile(b"")
