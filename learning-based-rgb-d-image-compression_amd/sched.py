"""Scheduling arithmetic of the engine pool.  No imports: bench.py loads this file by path before anything that could
start the HIP runtime (GPU_MAX_HW_QUEUES must be in the environment by then)."""


def balanced_workers(n_batches: int, max_workers: int = 20) -> int:
    """Engine instances for a job of n_batches: as few rounds as max_workers allows, every round full
    (20 batches -> 20 instances, one round; 48 -> 16, three rounds; a fixed 16 would leave 4 of 20 batches for a second
    round).  The cap is 20: with 24 instances of the c3 workload (145 GiB of workspaces) a lone instance's convolutions
    were measured a third slower (89 vs 67 ms per step) although the pooled job kept its rate."""
    n = max(1, int(n_batches))
    rounds = -(-n // max(1, max_workers))
    return -(-n // rounds)


MAX_SAFE_HW_QUEUES = 48


def check_hw_queues(env=None) -> int:
    """GPU_MAX_HW_QUEUES (the HIP runtime's count of hardware queues per process, fixed when it starts) inside the range this
    package has run cleanly with: 4 (the runtime's default) ... 48.  Round 4 measured 64 as the steadiest setting for 20 engine
    instances -- and with it kernel launches of OTHER libraries on their own streams (torch's, in the pipelined harness) began
    to fail with hipErrorInvalidValue ("CUDA driver error: 1", profiles/r04_harness_throughput.txt) once 24-32 worker streams
    held queues: every HIP stream of the process maps onto those queues and the failing launch was the first on a fresh
    stream, i.e. the process ran out of queue resources, not out of anything this package allocates.  Rather than sit next to
    that limit the pool and the harness REFUSE a larger value up front.  Returns the value in force (0: not set)."""
    import os

    raw = (env if env is not None else os.environ).get("GPU_MAX_HW_QUEUES", "")
    if not raw:
        return 0
    try:
        q = int(raw)
    except ValueError:
        raise ValueError(f"GPU_MAX_HW_QUEUES={raw!r} is not an integer")
    if q > MAX_SAFE_HW_QUEUES:
        raise ValueError(f"GPU_MAX_HW_QUEUES={q}: above {MAX_SAFE_HW_QUEUES} kernel launches on other streams of the process were "
                         "seen to fail with hipErrorInvalidValue on MI355X / ROCm 7.2 (DESIGN.md 3.3); use 24-48")
    return q
