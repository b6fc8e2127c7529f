import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import rgbd_amd
from rgbd_amd import ELIC_united, synth
H, W = 512, 640
net = ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
net.load_state_dict(synth.synthetic_state_dict(0, recipe=os.environ.get("RECIPE", "stress")))
net.update(force=True)
net = net.to("cuda")
r, d = synth.synthetic_batch(1, H, W, config_id=2)
rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
side = torch.cuda.Stream()
for name, ctx in (("default stream", None), ("side stream", side)):
    def run():
        te = td = 0.0
        for _ in range(3):
            out = net.compress(rgb, depth); net.decompress(out["r_strings"], out["d_strings"], out["shape"])
        torch.cuda.synchronize()
        N = 8
        for _ in range(N):
            t0 = time.perf_counter(); out = net.compress(rgb, depth); torch.cuda.synchronize(); t1 = time.perf_counter()
            net.decompress(out["r_strings"], out["d_strings"], out["shape"]); torch.cuda.synchronize(); t2 = time.perf_counter()
            te += t1 - t0; td += t2 - t1
        print(f"{name}: enc {te/N*1e3:.2f} ms dec {td/N*1e3:.2f} ms", flush=True)
    if ctx is None:
        run()
    else:
        with torch.cuda.stream(ctx):
            run()
