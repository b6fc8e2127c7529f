#!/usr/bin/env python3
"""HBM workspace of one engine instance after a compress() + decompress() of a shape (RGBD_NO_WS_REUSE=1: one-ended stack)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rgbd_amd  # noqa: E402
from rgbd_amd import ELIC_united, synth  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (4, 512, 640)
net = ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
net.load_state_dict(synth.synthetic_state_dict(0))
net.update(force=True)
net = net.to("cuda")
net.per_image_streams = True
r, d = synth.synthetic_batch(B, H, W, config_id=3)
rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
h = hashlib.sha256()
with torch.cuda.stream(torch.cuda.Stream()):
    for _ in range(3):  # eager, captured, replayed
        out = net.compress(rgb, depth)
        rec = net.decompress(out["r_strings"], out["d_strings"], out["shape"])
    torch.cuda.current_stream().synchronize()
    ws_codec = net.workspace_bytes()
    fw = net(rgb, depth)
torch.cuda.synchronize()
for s in out["r_strings"][0] + out["d_strings"][0] + out["r_strings"][1] + out["d_strings"][1]:
    h.update(s)
h.update(rec["x_hat"]["r"].cpu().numpy().tobytes())
h.update(rec["x_hat"]["d"].cpu().numpy().tobytes())
h.update(fw["x_hat"]["r"].cpu().numpy().tobytes())
print(f"{B}x{H}x{W}: workspace {ws_codec / 2**30:.3f} GiB after compress() + decompress(), {net.workspace_bytes() / 2**30:.3f} GiB after forward(); sha256(streams + x_hat + forward x_hat) {h.hexdigest()[:16]}")
