"""Stream-level pipelining of independent image groups on one GPU.

The entropy stage is a serial chain per stream (rANS state recurrence, 20 decode phases interleaved with the context
networks) that occupies a handful of CUs, while the transforms want the whole chip.  Image pairs are independent, so a
`CodecPool` keeps W engine instances (shared packed weights, own workspace, own HIP stream) and codes W groups of a batch
concurrently from W host threads: one group's serial coder phases overlap another group's convolutions.  Results are
identical to coding each group alone (the kernels are batch-invariant); only the wall clock changes.
"""
import os
import threading
from typing import List

import torch

from .elic_united import ELIC_united
from .sched import balanced_workers  # noqa: F401  (re-exported: rgbd_amd.pool.balanced_workers)


class CodecPool:
    """W engine instances of one model on one GPU.  `model_cls`: ELIC_united (default), STF_united, ELIC_united_R2D -- or the
    single-modal ELIC (models/elic.py), whose calls take one tensor: roundtrip(x) -> (outs, x_hat), batches = [(x,), ...].
    Use as a context manager, or call close(): the pool switches the DEVICE to blocking waits and close() switches it back."""

    def __init__(self, state_dict, config=None, workers: int = 2, device="cuda", per_image_streams: bool = True,
                 model_cls=ELIC_united, channel=None):
        self.single = getattr(model_cls, "_MODEL", "") == "ELIC"
        if channel is None:
            channel = 3 if self.single else 4
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.nets: List[ELIC_united] = []
        self.streams = []

        # torch hands out side streams from a pool of 32 per device: a 33rd instance would share its HIP stream with the
        # first, and one instance's graph capture would then swallow the other's work ("operation not permitted on an event
        # last recorded in a capturing stream")
        if workers > 32:
            raise ValueError(f"CodecPool: {workers} engine instances requested, at most 32 have a HIP stream of their own")
        from .sched import check_hw_queues

        check_hw_queues()  # (refuses a GPU_MAX_HW_QUEUES above the range launches were seen to survive)

        # Host threads that wait for this GPU sleep instead of spinning (hipDeviceScheduleBlockingSync): a pool keeps W
        # threads waiting on W streams.  Measured on c3 with 16 instances and HIP-graph launches: 15.1 -> 1.2 busy host
        # cores per rank at the same throughput (tools/host_cost.sh); RGBD_BLOCKING_SYNC=0 restores the spinning default.
        # The wait policy is a DEVICE flag of this process (hipSetDeviceFlags): it stays in force for every later wait on this
        # GPU, also outside the pool (a B = 1 latency pass in the same process runs under it: ~1 of 190 ms), until close().
        self._blocking_sync = False
        if workers > 1 and os.environ.get("RGBD_BLOCKING_SYNC", "1") != "0":
            from ._lib import set_blocking_sync

            torch.cuda.set_device(self.device)
            set_blocking_sync(True)
            self._blocking_sync = True
        for i in range(workers):
            if i == 0:
                net = model_cls(config=config, channel=channel).eval()  # ELIC_united, a variant of it, or the single-modal ELIC
                net.load_state_dict(state_dict)
                net.update(force=True)
                net = net.to(self.device)
            else:
                net = self.nets[0].clone_shared()  # same packed weights in HBM, own workspace
            net.per_image_streams = per_image_streams
            if workers >= 4:  # the chip is shared: tiles that cost the least CU time, not the ones that finish first alone
                net.set_tile_mode("throughput")
            self.nets.append(net)
            self.streams.append(torch.cuda.Stream(device=self.device))

    @property
    def workers(self):
        return len(self.nets)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _code(self, net, tensors):
        """compress() + decompress() of one group on one instance -> (compress output, tuple of reconstructions)."""
        out = net.compress(*tensors)
        if self.single:
            rec = net.decompress(out["strings"], out["shape"])
            return out, (rec["x_hat"],)
        rec = net.decompress(out["r_strings"], out["d_strings"], out["shape"])
        return out, (rec["x_hat"]["r"], rec["x_hat"]["d"])

    def close(self):
        """Drop the engine instances and let go of the sleeping wait policy (it returns to the runtime's default once no engine
        of this process is left: the policy never changes under a live engine)."""
        self.nets, self.streams = [], []
        if self._blocking_sync:
            from ._lib import set_blocking_sync

            torch.cuda.set_device(self.device)
            set_blocking_sync(False)  # (the instances dropped above are destroyed by now: under the policy they ran with)
            self._blocking_sync = False

    def _split(self, B):
        w = min(self.workers, B)
        base, rem = divmod(B, w)
        out, o = [], 0
        for i in range(w):
            n = base + (1 if i < rem else 0)
            out.append((o, o + n))
            o += n
        return out

    def _run(self, fn, n):
        res, err = [None] * n, [None] * n

        def work(i):
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.streams[i]):
                    res[i] = fn(i)
                self.streams[i].synchronize()
            except BaseException as e:  # surfaced to the caller below
                err[i] = e

        ts = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for e in err:
            if e is not None:
                raise e
        return res

    def roundtrip(self, *tensors: torch.Tensor):
        """compress() + decompress() of every group; returns (list of compress outputs, x_hat_r, x_hat_d) -- for the
        single-modal model: roundtrip(x) -> (list of compress outputs, x_hat)."""
        parts = self._split(tensors[0].shape[0])
        torch.cuda.current_stream().synchronize()

        def fn(i):
            a, b = parts[i]
            return self._code(self.nets[i], [t[a:b] for t in tensors])

        res = self._run(fn, len(parts))
        return ([r[0] for r in res],) + tuple(torch.cat([r[1][j] for r in res]) for j in range(len(res[0][1])))

    def roundtrip_many(self, batches, on_done=None):
        """Software pipeline over whole batches: the workers pull batches off a shared counter, so one batch's serial
        coder phases overlap the other workers' convolutions.  batches: list of (rgb, depth) -- (x,) for the single-modal
        model.  Returns [(compress_out, x_hat_r, x_hat_d)] ([(compress_out, x_hat)]) in batch order.

        Workers that start together stay in lock-step (same work, symmetric contention), so a job of K batches takes
        ceil(K / W) rounds and a last round with few workers leaves the chip idle: pick W so that the rounds are full
        (`balanced_workers`).  Breaking the lock-step with staggered starts or a cap on concurrent compress() calls was
        measured (tools/pool_sched_probe.py, c3, K = 20) and gains nothing: fewer instances in their transforms at a
        time lower the convolution throughput by as much as the overlap wins.  Round 3 tried two cohorts half a phase
        apart (the second half of the workers started 150 - 450 ms late, so that one cohort's coder phases fall into the
        other's transforms while each cohort keeps its lock-step): 19.8 / 20.0 / 18.5 / 17.6 Mpx/s against 20.4 without; and
        giving the first half (or quarter) of the instances high-priority streams, so that they run ahead without anybody
        starting late: 20.3-20.4 (20.9) Mpx/s against 21.2-21.4 on the same box."""
        # on_done(k, compress_out): called from the worker thread as soon as batch k is finished (bench.py hands the
        # finished streams to its gather thread there, so that the RCCL gather overlaps the other instances' work)
        n = len(batches)
        torch.cuda.current_stream().synchronize()
        W = min(self.workers, n)
        nxt, lock = [W], threading.Lock()

        def fn(i):
            outs = []
            k = i  # worker i starts with batch i (so W batches touch every instance once), then takes what is next
            while True:
                if k >= n:
                    return outs
                out, recs = self._code(self.nets[i], batches[k])
                outs.append((k, out) + recs)
                if on_done is not None:
                    on_done(k, out)
                with lock:
                    k = nxt[0]
                    nxt[0] += 1

        res = [None] * n
        for lst in self._run(fn, W):
            for item in lst:
                res[item[0]] = tuple(item[1:])
        return res

    def set_profile(self, on: bool):
        for n in self.nets:
            n.set_profile(on)

    def profile_read(self):
        tot = {"conv_ms": 0.0, "launches": 0, "flops": 0.0}
        for n in self.nets:
            p = n.profile_read()
            for k in tot:
                tot[k] += p[k]
        return tot
