"""Image-sharded data parallelism for the codec: one process per GPU, no collective on the data path.

Each image pair is coded independently, so rank r takes images r, r+world, ... and runs the whole
encode/decode locally on its own weight replica.  The only exchange is the gather of the finished per-image bitstreams
and metrics: an all_gather of int64 lengths followed by an all_gather of the zero-padded uint8 payload (RCCL when the
backend is "nccl", gloo on CPU for the tests).  The reference has no multi-GPU inference path at all.
"""
import os
from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend: str = None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RGBD_DIST_FORCE_INIT=1: build the process group even for a single rank, so that a one-GPU box can run every line of
    # the RCCL branch below (tests/test_gpu_distributed.py); a one-rank all_gather is a device copy
    force = os.environ.get("RGBD_DIST_FORCE_INIT") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:  # RGBD_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("RGBD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = min(local, torch.cuda.device_count() - 1)  # (only differs from LOCAL_RANK in such a rehearsal)
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard(n_items: int, rank: int, world: int) -> List[int]:
    """Indices of the items this rank owns (round-robin, so ranks stay balanced for any n)."""
    return list(range(rank, n_items, world))


def _comm_device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


class RankStreams(Sequence):
    """One rank's gathered byte strings: a view on the gathered payload, split into `bytes` objects only on access (a
    job that gathers hundreds of MB per step batch should not pay for thousands of copies it may never look at).  The
    payload may still live in this rank's HBM (where RCCL delivered it); it is copied to the host on first access."""

    def __init__(self, payload, lens: np.ndarray):
        self._p = payload  # numpy uint8 array, or a torch uint8 tensor (CPU or GPU)
        self._off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)

    def _host(self) -> np.ndarray:
        if not isinstance(self._p, np.ndarray):
            self._p = self._p.cpu().numpy()  # (drops the reference to the gathered device buffer)
        return self._p

    @property
    def on_device(self) -> bool:
        return not isinstance(self._p, np.ndarray) and self._p.is_cuda

    def __len__(self):
        return len(self._off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._host()[self._off[i]:self._off[i + 1]].tobytes()

    def __eq__(self, other):
        return list(self) == list(other)


def gather_streams(streams: Sequence[bytes], force: bool = False) -> List[Sequence[bytes]]:
    """All ranks receive every rank's list of byte strings (ranks may hold different numbers of strings): two small
    all_gathers (counts, lengths) and one all_gather of the zero-padded payload.  Element r of the result behaves like
    rank r's list of `bytes`.  With RCCL the payload is packed once into pinned host memory, uploaded asynchronously and
    gathered HBM to HBM over xGMI; the gathered bytes stay in HBM until somebody reads them (`RankStreams`): every result
    holds world x max_bytes of device memory until it is read (then the host copy replaces it) or dropped.
    `force`: run the collective path even in a one-rank group (test hook for one-GPU boxes)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return [list(streams)]
    world = dist.get_world_size()
    dev = _comm_device()
    on_gpu = dev.type == "cuda"
    total = sum(len(s) for s in streams)
    count = torch.tensor([len(streams), total], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = torch.stack(counts).cpu()
    max_n = max(int(counts[:, 0].max()), 1)
    max_b = max(int(counts[:, 1].max()), 1)
    lens = torch.zeros(max_n, dtype=torch.int64)
    if streams:
        lens[: len(streams)] = torch.tensor([len(s) for s in streams], dtype=torch.int64)
    payload = torch.empty(max_b, dtype=torch.uint8, pin_memory=on_gpu)  # one packing pass, straight into pinned memory
    view = payload.numpy()
    o = 0
    for s in streams:
        n = len(s)
        view[o:o + n] = np.frombuffer(s, dtype=np.uint8)
        o += n
    view[o:] = 0
    lens, payload = lens.to(dev, non_blocking=on_gpu), payload.to(dev, non_blocking=on_gpu)
    all_lens = [torch.empty_like(lens) for _ in range(world)]
    all_payload = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(all_lens, lens)
    dist.all_gather(all_payload, payload)
    if on_gpu:
        torch.cuda.current_stream().synchronize()  # the gather has landed in this rank's HBM
    out = []
    for r in range(world):
        n = int(counts[r][0])
        out.append(RankStreams(all_payload[r][: int(counts[r][1])], all_lens[r][:n].cpu().numpy()))
    return out


def gather_metrics(values: torch.Tensor, force: bool = False) -> torch.Tensor:
    """values: float64 [n_local, k] (same n_local on every rank) -> [world, n_local, k]."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return values.unsqueeze(0)
    dev = _comm_device()
    v = values.to(dev, torch.float64).contiguous()
    outs = [torch.zeros_like(v) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, v)
    return torch.stack([o.cpu() for o in outs])


def max_over_ranks(x: float, force: bool = False) -> float:
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(force: bool = False):
    if dist.is_initialized() and (dist.get_world_size() > 1 or force):
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
