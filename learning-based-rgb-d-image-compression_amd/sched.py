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
