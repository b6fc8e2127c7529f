#!/usr/bin/env python3
"""TesterUnited.test_model() on a folder of synthetic 480x640 RGB-D pairs (PNG in, container files out, metrics), one image
at a time like the reference (testing/tester_united.py:48-88) and with several images in flight (workers=W).
    python tools/harness_throughput.py [n_images] [W[:batch] ...]      (HARNESS_NO_PNG=1: skip the runs that save reconstructions)"""
import faulthandler
import logging
import os
import sys
import tempfile
import types

if os.environ.get("HARNESS_WATCHDOG"):  # all threads' stacks after that many seconds, then exit (a stuck pipeline shows where)
    faulthandler.dump_traceback_later(int(os.environ["HARNESS_WATCHDOG"]), exit=True)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

import rgbd_amd  # noqa: E402
from rgbd_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
BATCH = int(os.environ.get("HARNESS_BATCH", "4"))
Ws = [(int(v.split(":")[0]), int(v.split(":")[1]) if ":" in v else BATCH) for v in sys.argv[2:]] or [(1, 1), (4, BATCH), (8, BATCH)]
SAVES = (False, True) if not os.environ.get("HARNESS_NO_PNG") else (False,)
H, W_ = 480, 640
net = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
net.load_state_dict(synth.synthetic_state_dict(0))
net.update(force=True)
net = net.to("cuda")
logging.getLogger("test").setLevel(logging.WARNING)
with tempfile.TemporaryDirectory() as tmp:
    root = os.path.join(tmp, "nyu_test")
    os.makedirs(os.path.join(root, "rgb"))
    os.makedirs(os.path.join(root, "depth"))
    for i in range(n):
        r, d = synth.synthetic_pair(i, H, W_, config_id=3, smooth=True)
        Image.fromarray((r.transpose(1, 2, 0) * 255).astype(np.uint8)).save(os.path.join(root, "rgb", f"{i:04d}.png"))
        Image.fromarray((d[0] * 9000).astype(np.uint16)).save(os.path.join(root, "depth", f"{i:04d}.png"))
    cwd0 = os.getcwd()
    os.chdir(tmp)
    for save in SAVES:
        for w, BATCH in Ws:
            args = types.SimpleNamespace(channel=4, debug=False, experiment=f"exp{w}_{BATCH}_{int(save)}", dataset=root, model="ELIC_united",
                                         quality="2_2", checkpoint=None)
            t = rgbd_amd.TesterUnited(args, rgbd_amd.model_config(), net=net)
            t.save_reconstructions = save
            t.test_model(padding_mode="replicate0", padding=True, workers=w, batch=BATCH)  # warm-up: workspaces, clones
            rows, meters = t.test_model(padding_mode="replicate0", padding=True, workers=w, batch=BATCH)
            lat = sum(r["enc_time"] + r["dec_time"] for r in rows)
            print(f"workers {w:2d} batch {BATCH} save_png {int(save)}: job {t.job_mpx_per_s:6.2f} Mpx/s (wall, incl. file I/O + metrics) | "
                  f"reference metric sum(px)/sum(enc+dec) {n*H*W_/lat/1e6:6.2f} Mpx/s | avg enc {meters['avg_encode_time'].avg*1e3:7.1f} ms "
                  f"dec {meters['avg_deocde_time'].avg*1e3:7.1f} ms", flush=True)
            if w > 1 and os.environ.get("HARNESS_STAGES"):
                print("    host seconds per stage, all workers:", t.stage_seconds, "wall", round(n * H * W_ / t.job_mpx_per_s / 1e6, 3), flush=True)
    os.chdir(cwd0)  # (a profiler that finalises after us wants a working directory that still exists)
