"""The RCCL ("nccl") branch of rgbd_amd.distributed on hardware.

Two RCCL ranks cannot share a device, and the GPU box of the test tier has one: so (1) a ONE-rank nccl process group
(RGBD_DIST_FORCE_INIT=1) runs every line of the branch -- pinned packing, non-blocking upload, HBM -> HBM all_gather, the
lazy download in RankStreams, the reductions and the barrier -- on real codec streams, and (2) `bench.py --gpus 2` with the
nccl backend runs whenever two devices are visible (skipped otherwise; the round-end driver's 8-GPU tier runs the same
code).  The gloo twin of (2) is tests/test_gpu_bench.py::test_two_ranks_spawned_by_bench_itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT
from gpu_utils import require_gpu

pytestmark = pytest.mark.gpu

_ONE_RANK = r"""
import os, sys
sys.path.insert(0, os.environ["RGBD_ROOT"])
import torch, torch.distributed as dist
import rgbd_amd
from rgbd_amd import distributed, synth

rank, world, local = distributed.init_from_env()
assert (rank, world, local) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl"
net = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
net.load_state_dict(synth.synthetic_state_dict(0))
net.update(force=True)
net = net.to("cuda")
net.per_image_streams = True
r, d = synth.synthetic_batch(3, 128, 128, config_id=7)
out = net.compress(torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda())
mine = list(out["r_strings"][0]) + list(out["d_strings"][0]) + [b""]  # (an empty string is a legal element)
assert len(mine) == 7 and all(isinstance(s, bytes) for s in mine)
distributed.barrier(force=True)
got = distributed.gather_streams(mine, force=True)
assert len(got) == 1 and got[0].on_device, "the gathered payload should still be in HBM"
assert len(got[0]) == len(mine) and got[0][2] == mine[2] and got[0][-1] == b""
assert not got[0].on_device and list(got[0]) == mine
assert distributed.gather_streams([], force=True)[0] == []
m = distributed.gather_metrics(torch.arange(6, dtype=torch.float64).reshape(3, 2), force=True)
assert m.shape == (1, 3, 2) and m.sum().item() == 15.0
assert distributed.max_over_ranks(1.25, force=True) == 1.25
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", sum(len(s) for s in mine))
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _clean_env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                         "RGBD_DIST_BACKEND")}
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(kw)
    return e


@pytest.mark.timeout(600)
def test_rccl_branch_on_one_rank():
    require_gpu()
    env = _clean_env(RGBD_DIST_FORCE_INIT="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                     MASTER_PORT=str(_free_port()), RGBD_ROOT=ROOT)
    p = subprocess.run([sys.executable, "-c", _ONE_RANK], env=env, capture_output=True, text=True, timeout=540)
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


@pytest.mark.timeout(900)
def test_two_ranks_rccl_when_two_gpus_are_visible():
    require_gpu()
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: two RCCL ranks cannot share a device (the gloo twin runs in test_gpu_bench.py)")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--workers", "2"], env=_clean_env(), capture_output=True, text=True, timeout=840)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["value"] > 0.5 and r["scaling"] == "weak"


@pytest.mark.timeout(300)
def test_a_rank_without_a_device_of_its_own_refuses():
    """--gpus 2 on a one-GPU box with the nccl backend: the parent counts devices from sysfs without touching HIP and
    refuses; if the count cannot be read, the rank whose LOCAL_RANK has no device exits 2 and the job fails."""
    require_gpu()
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workers", "1", "--no-cpu-baseline", "--no-extras"], env=_clean_env(), capture_output=True,
                       text=True, timeout=280)
    assert p.returncode == 2, (p.returncode, p.stderr[-2000:])
    assert "GPU(s) visible" in p.stderr
