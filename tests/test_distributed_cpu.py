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
