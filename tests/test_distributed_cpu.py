"""The N>1 path on CPU: world_size-2 gloo processes shard the images and gather streams / metrics."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import rgbd_amd  # noqa: F401
    from rgbd_amd import distributed as D

    r, w, _ = D.init_from_env(backend="gloo")
    mine = D.shard(7, r, w)
    streams = [bytes([i]) * (10 + 3 * i) for i in mine]  # ragged, rank-dependent count
    allv = D.gather_streams(streams)
    metrics = D.gather_metrics(torch.tensor([[float(r), 2.0 * r]], dtype=torch.float64))
    mx = D.max_over_ranks(1.0 + r)
    empty = D.gather_streams([] if r == 0 else [b"xyz"])
    D.barrier()
    q.put((r, mine, [[len(s) for s in lst] for lst in allv], [[s[:1] for s in lst] for lst in allv], metrics.tolist(), mx,
           [[bytes(s) for s in lst] for lst in empty]))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=100) for _ in ps)
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    assert got[0][1] == [0, 2, 4, 6] and got[1][1] == [1, 3, 5]
    for r, _, lens, heads, metrics, mx, empty in got:
        assert lens == [[10, 16, 22, 28], [13, 19, 25]]
        assert heads == [[b"\x00", b"\x02", b"\x04", b"\x06"], [b"\x01", b"\x03", b"\x05"]]
        assert metrics == [[[0.0, 0.0]], [[1.0, 2.0]]]
        assert mx == 2.0
        assert empty == [[], [b"xyz"]]


def test_single_process_passthrough():
    import rgbd_amd  # noqa: F401
    from rgbd_amd import distributed as D

    assert D.gather_streams([b"ab", b""]) == [[b"ab", b""]]
    assert D.shard(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.max_over_ranks(3.5) == 3.5


def _bench(*argv, env=None, timeout=150):
    import subprocess
    import sys

    from conftest import ROOT

    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.timeout(200)
def test_bench_spawns_its_own_ranks():
    """`bench.py --gpus 2` with no launcher around it starts 2 ranks itself (one process per GPU; here a gloo rehearsal of
    the plumbing: rendezvous, barrier, max-over-ranks timing, the stream all-gather) and prints rank 0's one JSON line."""
    import json

    p = _bench("--gpus", "2", "--steps", "3", "--rehearse")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["gathered_ok"] is True
    assert rec["streams_per_rank"] == [4, 5]


@pytest.mark.timeout(200)
def test_bench_rank_count_mismatch_is_an_error():
    # started as one of 3 ranks but told --gpus 2: refuse instead of reporting a wrong n_gpus
    p = _bench("--gpus", "2", "--rehearse", env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and "does not match" in p.stderr
    # a failing rank takes the whole job down with a non-zero exit code
    p = _bench("--gpus", "2", "--rehearse", "--steps", "1", env={"RGBD_REHEARSE_FAIL_RANK": "1"})
    assert p.returncode != 0


def test_launcher_counts_gpus_from_sysfs_without_torch(tmp_path, monkeypatch):
    """bench.visible_gpu_count: KFD topology nodes with SIMDs whose render node this process may open, narrowed by
    *_VISIBLE_DEVICES -- what the launcher parent uses instead of torch.cuda.device_count()."""
    import importlib.util

    from conftest import ROOT

    spec = importlib.util.spec_from_file_location("_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    base, dri = tmp_path / "nodes", tmp_path / "dri"
    dri.mkdir()
    for i, (simd, minor, present) in enumerate([(0, 0, False), (1024, 128, True), (1024, 129, True), (1024, 130, False)]):
        (base / str(i)).mkdir(parents=True)
        (base / str(i) / "properties").write_text(f"cpu_cores_count {8 if not simd else 0}\nsimd_count {simd}\n"
                                                  f"drm_render_minor {minor}\n")
        if present:
            (dri / f"renderD{minor}").write_text("")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count(str(base), str(dri)) == 2  # node 0 is a CPU, node 3's render node is not ours
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert bench.visible_gpu_count(str(base), str(dri)) == 1
    assert bench.visible_gpu_count(str(tmp_path / "missing"), str(dri)) is None


@pytest.mark.timeout(200)
def test_launcher_parent_stays_off_torch():
    """The parent of a self-spawned job must not have loaded torch / HIP when it starts its ranks (bench.py asserts it):
    run the launcher with an import hook that fails the process if the parent imports torch."""
    import subprocess
    import sys

    from conftest import ROOT

    code = ("import sys, runpy\n"
            "class Block:\n"
            "    def find_spec(self, name, path=None, target=None):\n"
            "        if name == 'torch':\n"
            "            raise ImportError('launcher parent imported torch')\n"
            "sys.meta_path.insert(0, Block())\n"
            f"sys.argv = [{os.path.join(ROOT, 'bench.py')!r}, '--gpus', '2', '--steps', '1', '--rehearse']\n"
            f"runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n")
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stderr[-2000:]
    assert any(ln.startswith("{") for ln in p.stdout.splitlines())
