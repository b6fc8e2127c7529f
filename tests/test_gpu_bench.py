"""bench.py on the GPU box: the one-line JSON contract the driver parses (metric / value / roofline / cpu_baseline /
latency), and the N-rank path end to end -- two ranks started by bench.py itself, sharing this box's one GPU, with gloo
standing in for RCCL (two RCCL ranks cannot share a device)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from gpu_utils import require_gpu

pytestmark = pytest.mark.gpu


def _run(*argv, env=None, timeout=600):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_single_rank_line_has_the_contract_fields():
    require_gpu()
    r = _run("--steps", "8", "--warmup", "2", "--workers", "2")  # c3 codes four steps per engine call: two calls of 16 images
    assert r["metric"].startswith("RGB-D Mpixels/s") and r["unit"] == "Mpx/s" and r["n_gpus"] == 1
    assert r["steps"] == 8 and r["warmup"] == 2 and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["dtype"] == "f32" and r["data"] == "synthetic" and r["vs_baseline"] is None
    assert r["config"]["workload"] == "c3_4x480x640" and r["config"]["image"] == [480, 640]
    assert r["config"]["steps_per_call"] == 4 and r["config"]["images_per_call"] == 16 and r["config"]["calls"] == 2
    assert r["value"] > 1.0 and abs(r["value"] - 4 * 480 * 640 * 8 / (r["ms_per_step"] * 8e-3) / 1e6) < 0.02 * r["value"]
    rf = r["roofline"]
    assert rf["bound"] == "mfma" and rf["peak"] == 157.3 and rf["unit"] == "TFLOP/s"
    assert 0.05 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert 0.3 < rf["isolated"]["frac"] < 1.0 and 300 <= rf["launches_per_call"] <= 360 and abs(rf["launches_per_step"] * 4 - rf["launches_per_call"]) < 0.1  # (333 before reference arithmetic: gather / scatter launches of the stride-2 deconv recipes on top) 594 -> 506 with fused block tails, -> 329 with RGB / depth layer pairs as one launch, + 4: the 192-channel slice's local-context convs as two half-tap launches
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mpx/s" and cb["cores"] >= 1 and cb["value"] > 0
    lat = r["latency"]  # the reference tester's metric: B = 1, synchronised windows
    assert lat["batch"] == 1 and lat["engine_instances"] == 1 and lat["value"] > 0.2
    assert abs(lat["value"] - 480 * 640 / ((lat["enc_ms_per_image"] + lat["dec_ms_per_image"]) * 1e-3) / 1e6) < 0.02 * lat["value"]
    assert r["latency_trained_like"]["value"] > lat["value"] * 0.8
    assert r["latency_high_rate"]["batch"] == 1 and r["latency_high_rate"]["value"] > 0.2  # symbols on CDF rows of 300 ... 3000 entries
    assert r["vs_cpu"]["throughput"] > r["vs_cpu"]["latency_tester_semantics"] > 1.0
    assert [w["workload"] for w in r["workloads"]] == ["c3_4x480x640", "c2_8x256x256", "c5_stf_1x512x512", "c5_stf_4x512x512"]
    for w in r["workloads"][2:]:  # BASELINE config 5 rides along with its own roofline and CPU baseline
        assert w["value"] > 1.0 and 0.05 < w["roofline"]["frac"] < 1.0 and w["cpu_baseline"]["kind"] == "port" and w["vs_cpu"] > 1.0
    # what the line costs and what it was measured under (round-3 review)
    assert r["config"]["pairs_in_flight"] == 32 and 0.1 < r["config"]["hbm_workspace_gib_per_instance"] <= 12.0
    assert abs(r["config"]["hbm_workspace_gib"] - 2 * r["config"]["hbm_workspace_gib_per_instance"]) < 0.05
    sus = r["sustained"]
    assert sus["steps"] == 24 and len(sus["ms_per_step_by_round"]) == 3 and sus["value"] > 0.5 * r["value"]
    par = r["parity"]
    assert par["of"] == 21 and 0 < par["goldens_identical"] <= 21 and "f_480x640_stress" in par["operating_point"]
    assert "BATCH THROUGHPUT" in r["vs_cpu"]["note"]


@pytest.mark.timeout(900)
def test_two_ranks_spawned_by_bench_itself():
    require_gpu()
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "2", "--workers", "2", env={"RGBD_DIST_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["config"]["images_per_gpu"] == 4
    assert r["value"] > 0.5 and "latency" not in r and "cpu_baseline" not in r
