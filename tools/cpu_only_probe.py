import sys, os, time, json
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from rgbd_amd import synth
sd = synth.synthetic_state_dict(0)
r = b.cpu_baseline(sd, 480, 640, 3, "ELIC_united", seconds_budget=20.0, batch8=False)
print("CPU-ONLY", json.dumps(r["legs"]))
