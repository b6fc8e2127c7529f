"""Step-logged probe of the pooled HIP-graph replay path (two engine instances, two alternating call shapes, every result
against a single-instance reference), with a faulthandler watchdog: writes gpurun_out/hang_probe.log.  Used to localise a
hang in the runtime; see DESIGN.md 3.5."""
import faulthandler, sys, os, time
sys.path.insert(0, "/root/repo")
log = open("/root/repo/gpurun_out/hang_probe.log", "w")
faulthandler.dump_traceback_later(70, exit=True, file=log)
def P(*a):
    print(time.strftime("%H:%M:%S"), *a, file=log, flush=True)
import torch
import rgbd_amd
from rgbd_amd import synth
sd = synth.synthetic_state_dict(0)
P("start")
net = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
net.load_state_dict(sd); net.update(force=True); net = net.to("cuda")
pool = rgbd_amd.CodecPool(sd, config=rgbd_amd.model_config(), workers=2, device="cuda", per_image_streams=True)
P("pool built")
r, d = synth.synthetic_batch(4, 128, 128, config_id=11)
rgb, depth = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
net.per_image_streams = True
ref = net.compress(rgb, depth); ref_rec = net.decompress(ref["r_strings"], ref["d_strings"], ref["shape"])
P("ref done")
outs, xr, xd = pool.roundtrip(rgb, depth)
P("roundtrip done")
many = pool.roundtrip_many([(rgb, depth)] * 3)
P("many3 done")
r2, d2 = synth.synthetic_batch(2, 128, 192, config_id=12)
rgb2, depth2 = torch.from_numpy(r2).cuda(), torch.from_numpy(d2).cuda()
ref2 = net.compress(rgb2, depth2); ref2_rec = net.decompress(ref2["r_strings"], ref2["d_strings"], ref2["shape"])
P("ref2 done")
seq = [(rgb, depth), (rgb2, depth2)] * 6
res = pool.roundtrip_many(seq)
P("seq done")
for k, (out, mxr, mxd) in enumerate(res):
    want, want_rec = (ref, ref_rec) if k % 2 == 0 else (ref2, ref2_rec)
    P(k, out["r_strings"] == want["r_strings"], bool(torch.equal(mxr, want_rec["x_hat"]["r"])))
P("graphs", [n.graph_count() for n in pool.nets])
