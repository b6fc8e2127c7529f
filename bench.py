#!/usr/bin/env python3
"""Headline benchmark: RGB-D Mpixels/s, encode+decode, ELIC_united q=2_2 on MI355X (BASELINE.json).

    python bench.py                                   # 1 GPU, c3 (4 x 480x640 per step) + c2 + B=1 latency + CPU baseline
    python bench.py --gpus N --steps K --warmup W     # N>1: starts N ranks itself (one process per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W         # ... or is started as one of N ranks (RANK/WORLD_SIZE in the env)

One step = compress() + decompress() of one batch of synthetic RGB-D pairs per rank.  The default workload is BASELINE
config 3's per-GPU share: 4 pairs of 480x640 (replicate-padded to 512x640) per GPU and step, so `--gpus 8` codes config
3's 32 images per step (weak scaling: every rank codes its own images; no collective on the data path, only the
all-gather of the finished streams).  An engine call codes `--steps-per-call` steps together (default 4 for c3 and c2: the
driver's 20 steps are 5 calls of 16 images on 5 engine instances -- same images in flight as 20 instances of 4, faster
kernels; streams are per image and the kernels batch-invariant, so no output bit depends on it).  Inputs are resident in
HBM before the timed region; conv profiling (HIP events) runs in a separate pass after it.  Rank 0 prints ONE JSON line:

  value / ms_per_step   wall-clock job throughput with W engine instances in flight per GPU (`engine_instances`)
  latency               the reference tester's calling pattern (testing/tester_united.py:142-147,180-186): B=1, one engine
                        instance, device-synchronised windows around compress() and decompress(), Mpx/s = sum(px) /
                        (sum(enc) + sum(dec))  -- SURVEY.md 8(d)'s metric definition
  cpu_baseline          the CPU oracle at the same B=1 semantics on this box's host cores; `vs_cpu` holds both ratios
  workloads             the same throughput figure for the secondary workload (c2: 8 x 256x256)
  roofline              conv kernel (fp32 MFMA): job-level achieved FLOP/s (algorithmic FLOPs of the reference's layers; `executed_frac`
                        = what the launches compute), `isolated` = one engine instance alone, `hbm` = PMC bytes of the conv launches,
                        `entropy` = the entropy stage (rANS ns per symbol on the model's own symbols, checkerboard pass GB/s)
  sustained             three more rounds on the same engine instances right after the headline (60 steps at --steps 20)
  parity                the parity state the numbers were measured under (tests/golden/parity_floors.json: how many reference
                        goldens are bit-identical, the largest dbpp / dPSNR, the state at the bench's own operating point)
  config.hbm_workspace_gib / pairs_in_flight   what the throughput costs in HBM and in images held at once
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per engine instance (the HIP default of 4 makes streams share queues, and a long serial coder
# kernel then blocks another instance's convolutions); must be set before the HIP runtime initialises
_USER_QUEUES = "GPU_MAX_HW_QUEUES" in os.environ
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # refined in main() once --workers is known (still before HIP starts)

WORKLOADS = {
    # name: (batch per GPU, H, W, synthetic config id, model)
    "c2_8x256x256": (8, 256, 256, 2, "ELIC_united"),
    "c3_4x480x640": (4, 480, 640, 3, "ELIC_united"),   # BASELINE config 3: 32 images over 8 GPUs = 4 per GPU
    "c3_8x480x640": (8, 480, 640, 3, "ELIC_united"),   # (W, B) sweep at constant pairs in flight (round-4 review, item 5):
    "c3_16x480x640": (16, 480, 640, 3, "ELIC_united"),  # two / four / eight of c3's steps per engine call
    "c3_32x480x640": (32, 480, 640, 3, "ELIC_united"),
    "c5_stf_1x512x512": (1, 512, 512, 5, "STF_united"),  # BASELINE config 5 (Swin transforms)
    "c5_stf_4x512x512": (4, 512, 512, 5, "STF_united"),  # ... four pairs per step: the step is not one serial coder chain
}
# steps coded per engine call by default (--steps-per-call 0): the (instances x batch) sweep at 80 pairs in flight,
# profiles/r05_call_batch_sweep.txt -- 20 x 4, 10 x 8, 5 x 16, 3 x 32 images: 5-6 instances of 16 are the best point
DEFAULT_STEPS_PER_CALL = {"c3_4x480x640": 4, "c2_8x256x256": 4}  # (c2, same box: 20 x 8 28.0 ms, 10 x 16 26.1, 5 x 32 25.4-25.6)
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table (dense f32 matrix)


def host_cpu_info():
    """CPU model, logical CPUs this process may run on and the physical cores among them (from /proc/cpuinfo)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    model, cores, cur = "unknown", set(), {}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "processor":
                    cur = {"cpu": int(v)}
                elif k == "model name":
                    model = v
                elif k in ("physical id", "core id"):
                    cur[k] = v
                    if "physical id" in cur and "core id" in cur and cur.get("cpu") in allowed:
                        cores.add((cur["physical id"], cur["core id"]))
    except OSError:
        pass
    quota = None
    try:  # cgroup v2 CPU quota of the box, if any ("max 100000" = none)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    return {"model": model, "logical_allowed": len(allowed), "physical_allowed": len(cores) or len(allowed), "cgroup_cpu_quota": quota}


def cpu_baseline(sd, H, W, cid, model="ELIC_united", seconds_budget=25.0, batch8=True):
    """The oracle (CPU restatement of the reference path: PyTorch-CPU eager + C coder) timed on this host on pairs of the
    workload's image size (replicate-padded to multiples of 64 like the harness does), SURVEY 8(d): B = 1 (tester
    semantics) with 16 threads (the figure of rounds 1-2) and with one thread per physical core this process may use
    (capped by a cgroup CPU quota), and one B = 8 batch with the better of the two.  `value` is the best of all legs."""
    import torch

    from oracle import elic_oracle as eo
    from rgbd_amd import synth

    info = host_cpu_info()
    phys = info["physical_allowed"]
    if info["cgroup_cpu_quota"]:
        phys = max(1, min(phys, int(info["cgroup_cpu_quota"] + 0.5)))
    orc = eo.OracleCodec(sd) if model == "ELIC_united" else eo.oracle_stf(sd)
    orc.update()

    def leg(threads, B, budget, max_runs):
        torch.set_num_threads(threads)
        done, spent, best, runs = 0, 0.0, None, []
        while spent < budget and done < max_runs:
            r, d = synth.synthetic_batch(B, H, W, config_id=cid, start=done * B)
            r, d = eo.pad_replicate0(torch.from_numpy(r)), eo.pad_replicate0(torch.from_numpy(d))
            t0 = time.time()
            out = orc.compress(r, d)
            orc.decompress(out["r_strings"], out["d_strings"], out["shape"])
            dt = time.time() - t0
            spent += dt
            done += 1
            best = dt if best is None else min(best, dt)
            runs.append(round(B * H * W / dt / 1e6, 5))
        # `value`: the fastest run (the CPU's best case); `runs_mpx_s`: every run, so that the spread is on the line
        return {"threads": threads, "batch": B, "runs": done, "value": round(B * H * W / best / 1e6, 5), "runs_mpx_s": runs}

    legs = [leg(min(16, info["logical_allowed"]), 1, seconds_budget * 0.6, 6)]
    if phys != legs[0]["threads"]:
        legs.append(leg(phys, 1, seconds_budget * 0.4, 4))
    top = max(legs, key=lambda x: x["value"])
    if batch8:
        legs.append(leg(top["threads"], 8, 1.0, 1))  # one batch of 8 (the reference's interleaved B > 1 stream format)
    best = max(legs, key=lambda x: x["value"])
    return {"value": best["value"], "unit": "Mpx/s", "cores": best["threads"], "kind": "port", "cpu": info, "legs": legs,
            "value_16_threads_b1": legs[0]["value"],
            "sample": f"best leg: {best['runs']} x (B={best['batch']}, {H}x{W}) enc+dec with {best['threads']} torch-CPU "
                      f"threads; legs = 16 threads / one thread per usable physical core at B=1 (tester semantics), then B=8"}


def parity_state():
    """The parity state the numbers of this line were measured under: tests/golden/parity_floors.json (recorded on MI355X by
    tests/test_gpu_parity_pinned.py against the reference's golden streams; tests/golden/update_floors.py)."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "parity_floors.json")) as f:
            sm = json.load(f)["_summary"]
        return {k: sm[k] for k in ("goldens_identical", "of", "bpp_identical", "dpsnr_within_1e-4", "max_dbpp", "max_dpsnr",
                                   "operating_point", "forced_context") if k in sm}
    except (OSError, KeyError, ValueError):
        return None


def visible_gpu_count(base="/sys/class/kfd/kfd/topology/nodes", dri="/dev/dri"):
    """GPUs this process could open, counted WITHOUT loading torch or the HIP runtime (the launcher parent must never
    initialise the GPU: its children are fork+exec'd): KFD topology nodes with SIMDs whose DRM render node is accessible,
    narrowed by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  None when the topology cannot be read
    (the ranks then check for themselves and exit 2 when LOCAL_RANK has no device)."""
    try:
        nodes = sorted(os.listdir(base), key=lambda x: int(x) if x.isdigit() else 1 << 30)
    except OSError:
        return None
    n = 0
    for node in nodes:
        props = {}
        try:
            with open(os.path.join(base, node, "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
        except OSError:
            continue
        if int(props.get("simd_count", "0") or 0) <= 0:
            continue  # a CPU node
        minor = props.get("drm_render_minor")
        if minor not in (None, "", "0") and not os.access(os.path.join(dri, f"renderD{minor}"), os.R_OK | os.W_OK):
            continue  # listed by the host's topology but not handed to this container
        n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(n, argv):
    """`--gpus N` without a launcher: start N ranks (one process per GPU) from this process, which has not touched the GPU
    and never does; relay rank 0's JSON line (the children share our stdout) and fail if any rank fails."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RGBD_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc, alive = 0, set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for k in alive:  # exactly the processes started above
                    procs[k].terminate()
        time.sleep(0.05)
    return rc


def rehearse(args):
    """CPU rehearsal of the N-rank plumbing (tests/test_distributed_cpu.py): rendezvous, barrier / max-over-ranks timing
    and the stream all-gather with gloo, on made-up byte strings instead of the codec's -- no GPU, no library."""
    import hashlib

    from rgbd_amd import distributed

    rank, world, _ = distributed.init_from_env(backend="gloo")
    assert world == args.gpus, (world, args.gpus)
    if os.environ.get("RGBD_REHEARSE_FAIL_RANK") == str(rank):  # test hook: a rank that dies before the first collective
        sys.exit(3)
    streams = [hashlib.sha256(f"{rank}:{i}".encode()).digest() * (1 + (rank + i) % 3) for i in range(4 + rank)]
    distributed.barrier()
    t0 = time.perf_counter()
    got = None
    for _ in range(max(args.steps, 1)):
        got = distributed.gather_streams(streams)
    distributed.barrier()
    elapsed = distributed.max_over_ranks(time.perf_counter() - t0)
    ok = all(list(got[r]) == [hashlib.sha256(f"{r}:{i}".encode()).digest() * (1 + (r + i) % 3) for i in range(4 + r)]
             for r in range(world))
    if rank == 0:
        print(json.dumps({"metric": "rehearsal", "n_gpus": world, "steps": args.steps, "gathered_ok": ok,
                          "streams_per_rank": [len(got[r]) for r in range(world)], "elapsed_s": round(elapsed, 4)}), flush=True)
    import torch.distributed as dist

    if dist.is_initialized():
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="c3_4x480x640", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of CPU-oracle work for cpu_baseline (the B = 8 leg "
                    "runs only with >= 20)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workload and the B=1 latency pass")
    ap.add_argument("--workers", type=int, default=0,
                    help="engine instances (HIP streams) per GPU; 1 = no overlap; 0 = as few full rounds of at most 20 "
                         "instances as --steps allows (20 steps -> 20 instances, 48 -> 16)")
    ap.add_argument("--steps-per-call", type=int, default=0,
                    help="steps one engine call codes together (0 = the workload's measured best: 4 for c3 and c2, else 1).  A step stays "
                         "one batch of the workload; G steps per call means calls of G x that batch on fewer engine "
                         "instances at the same number of pairs in flight (profiles/r05_call_batch_sweep.txt).  Kernels are "
                         "batch-invariant and streams are per image, so no output bit depends on it; reduced to a divisor of --steps")
    ap.add_argument("--tile-mode", default="auto", choices=["auto", "latency", "throughput"],
                    help="conv tile tables: auto = throughput tiles when >= 4 engine instances share the GPU (CodecPool's rule)")
    ap.add_argument("--rehearse", action="store_true", help=argparse.SUPPRESS)  # CPU test of the N-rank plumbing
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher around us: become the launcher.  Nothing in this process has touched (or will touch) the GPU.
        # (no torch / HIP call here: the GPUs are counted from sysfs; a rank without a device exits 2 by itself)
        if not args.rehearse and os.environ.get("RGBD_DIST_BACKEND") != "gloo":
            have = visible_gpu_count()
            if have is not None and have < args.gpus:
                print(f"[bench] --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
                sys.exit(2)
        assert "torch" not in sys.modules, "the launcher parent must not load torch / HIP before it starts its ranks"
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        print(f"[bench] WORLD_SIZE={env_world} does not match --gpus {args.gpus}", file=sys.stderr)
        sys.exit(2)
    if args.rehearse:
        sys.exit(rehearse(args))
    G = args.steps_per_call if args.steps_per_call > 0 else DEFAULT_STEPS_PER_CALL.get(args.workload, 1)
    G = max(1, min(G, args.steps))
    while args.steps % G:
        G -= 1
    calls = args.steps // G
    if args.workers <= 0:
        # rgbd_amd.pool.balanced_workers, loaded by file: nothing may load torch / HIP before GPU_MAX_HW_QUEUES is final
        import importlib.util

        spec = importlib.util.spec_from_file_location(
            "_rgbd_sched", os.path.join(ROOT, "learning-based-rgb-d-image-compression_amd", "sched.py"))
        sched = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sched)
        args.workers = sched.balanced_workers(calls)
    if not _USER_QUEUES:
        # hardware queues of this process (fixed when HIP starts).  ELIC_united with 20 instances, round-4 sweep of the driver's
        # command: 16 / 24 / 28 / 32 / 40 / 48 / 64 queues -> 58.8 / 56.9 / 56.3 / 56.0 / 55.9 / 56.4 / 55.6 ms per step (streams
        # that share a queue serialise).  40, not 64: with 64 a torch kernel launch of the pipelined harness failed ("CUDA
        # driver error: 1"; 48 and 40 are clean there) -- no need to sit next to that limit for 0.5 %.  STF_united with 16
        # instances wants few (24: 30.0-30.6 vs 28.3-30.4 Mpx/s with 64)
        os.environ["GPU_MAX_HW_QUEUES"] = str(40 if WORKLOADS[args.workload][4] == "ELIC_united" else max(24, args.workers + 8))

    import torch

    import rgbd_amd
    from rgbd_amd import CodecPool, distributed, synth

    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("RGBD_DIST_BACKEND") != "gloo":
        have, want = torch.cuda.device_count(), int(os.environ.get("LOCAL_RANK", "0"))
        if want >= have:  # one process per GPU: a rank without a device of its own must not double up on another rank's
            print(f"[bench] rank {os.environ.get('RANK')}: LOCAL_RANK {want} but {have} GPU(s) visible", file=sys.stderr)
            sys.exit(2)
    rank, world, local = distributed.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    B, H, W, cid, model = WORKLOADS[args.workload]
    recipe = os.environ.get("RGBD_BENCH_RECIPE", "stress")  # experiments only: the headline runs the stress recipe
    sd = synth.synthetic_state_dict(0, model=model) if recipe == "stress" else synth.synthetic_state_dict(0, model=model, recipe=recipe)
    # per-image stream sets (the unit that shards across GPUs); W engine instances overlap one group's serial coder
    # phases with another group's convolutions
    net = CodecPool(sd, config=rgbd_amd.model_config(), workers=args.workers, device=dev, per_image_streams=True,
                    model_cls=rgbd_amd.modelZoo[model])
    if args.tile_mode != "auto":
        for n_ in net.nets:
            n_.set_tile_mode(args.tile_mode)
    tile_mode = args.tile_mode if args.tile_mode != "auto" else ("throughput" if args.workers >= 4 else "latency")

    def make_inputs(Bq, Hq, Wq, cidq):
        r, d = synth.synthetic_batch(Bq, Hq, Wq, config_id=cidq, start=rank * Bq)
        rgb, depth = torch.from_numpy(r).to(dev), torch.from_numpy(d).to(dev)
        ph, pw = (-Hq) % 64, (-Wq) % 64
        if ph or pw:  # dataset/utils.py:58-67 "replicate0"
            rgb = torch.nn.functional.pad(rgb, (0, pw, 0, ph), mode="replicate")
            depth = torch.nn.functional.pad(depth, (0, pw, 0, ph), mode="replicate")
        return rgb.contiguous(), depth.contiguous(), (Hq + ph, Wq + pw)

    host = {}

    def timed(pool, rgb, depth, ncalls, nwarm, rounds=1, per_call=1):
        """ncalls engine calls (each codes `per_call` steps' worth of images), nwarm untimed warm-up STEPS first"""
        def run(k):
            # every step codes one full batch (compress + decompress); the W engine instances keep W steps in flight, so
            # one step's serial coder phases overlap another step's convolutions.  All k steps finish before this returns.
            if world == 1:
                return pool.roundtrip_many([(rgb, depth)] * k)
            # N > 1: the job's only exchange is the all-gather of the finished streams (RCCL).  A gather thread takes the steps
            # in index order as the instances finish them -- every rank issues the same sequence of collectives -- so a
            # step's gather overlaps the other instances' work instead of sitting behind the whole round (round-3 review);
            # it keeps only the last result (each holds world x largest-rank bytes of HBM until read or dropped).
            import threading

            done = [threading.Event() for _ in range(k)]
            outs = [None] * k
            err = []

            def on_done(i, out):
                outs[i] = out
                done[i].set()

            def gatherer():
                try:
                    torch.cuda.set_device(dev)
                    with torch.cuda.stream(gather_stream):
                        for i in range(k):
                            while not done[i].wait(0.5):
                                if err:
                                    return
                            distributed.gather_streams(outs[i]["r_strings"][0] + outs[i]["d_strings"][0])
                except BaseException as e:  # surfaced below
                    err.append(e)

            t = threading.Thread(target=gatherer)
            t.start()
            try:
                res = pool.roundtrip_many([(rgb, depth)] * k, on_done=on_done)
            except BaseException as e:
                err.append(e)
                raise
            finally:
                t.join()
            if err:
                raise err[0]
            return res

        if nwarm:
            # every engine instance sizes its workspace on its first batch and captures its HIP graphs on the second: the
            # timed steps then run the way a long job runs (RGBD_BENCH_WARM_ROUNDS=1: time the capturing calls instead)
            w = min(pool.workers, ncalls)
            wcalls = max(-(-nwarm // per_call), int(os.environ.get("RGBD_BENCH_WARM_ROUNDS", "2")) * w)
            host["warmup_steps_run"] = wcalls * per_call
            run(wcalls)
        out = []
        for _ in range(rounds):
            distributed.barrier()
            torch.cuda.synchronize()
            c0 = os.times()
            t0 = time.perf_counter()
            res = run(ncalls)
            torch.cuda.synchronize()
            distributed.barrier()
            dt = time.perf_counter() - t0
            c1 = os.times()
            host["cores_busy"] = round(((c1.user - c0.user) + (c1.system - c0.system)) / dt, 2)  # this rank's host threads
            out.append((distributed.max_over_ranks(dt), res[-1][0]))
        return out[0] if rounds == 1 else out

    gather_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    rgb, depth, padded = make_inputs(B * G, H, W, cid)  # one engine call codes G steps (G x B distinct images)
    elapsed, last = timed(net, rgb, depth, calls, args.warmup, per_call=G)
    host_cores = host.get("cores_busy")
    # sustained: three more rounds of --steps steps on the same instances, timed as one region (what a long job settles at)
    sustained = None
    if not args.no_extras:
        sus = timed(net, rgb, depth, calls, 0, rounds=3, per_call=G)
        t_sus = sum(e for e, _ in sus)
        sustained = {"steps": 3 * args.steps, "ms_per_step": round(t_sus / (3 * args.steps) * 1e3, 3),
                     "value": round(world * B * H * W * 3 * args.steps / t_sus / 1e6, 4), "unit": "Mpx/s",
                     "ms_per_step_by_round": [round(e / args.steps * 1e3, 3) for e, _ in sus]}
    workspace_gib = round(sum(n_.workspace_bytes() for n_ in net.nets) / 2 ** 30, 2)
    if world > 1:  # outside the timed region: what every rank received from this rank is what this rank sent
        mine = [s for s in last["r_strings"][0] + last["d_strings"][0]]
        got = distributed.gather_streams(mine)
        assert len(got) == world and list(got[rank]) == mine, "stream gather returned different bytes"

    # ---- conv profile, separate passes (not in the timed region): one engine instance, nothing else on the chip.  Twice:
    # with the tiles the timed region ran (throughput tiles when the chip is shared) and with the latency tiles, which
    # is what a lone engine instance runs by default (tile choice never changes a result bit, only the speed).
    solo = net.nets[0]

    def conv_pass(mode, one=None, xr=None, xd=None):
        one, xr, xd = one or solo, rgb if xr is None else xr, depth if xd is None else xd
        one.set_tile_mode(mode)
        one.set_profile(True)
        for _ in range(2):
            o = one.compress(xr, xd)
            one.decompress(o["r_strings"], o["d_strings"], o["shape"])
        p = one.profile_read()
        one.set_profile(False)
        return p

    prof1 = conv_pass(tile_mode)
    prof_lat = prof1 if tile_mode == "latency" else conv_pass("latency")
    if os.environ.get("RGBD_BENCH_DEBUG"):
        print("[bench] conv passes:", tile_mode, prof1, "| latency", prof_lat, file=sys.stderr)
    solo.set_tile_mode(tile_mode)
    flops_step = prof1["flops"] / 2.0 / G        # (the pass runs two calls of G steps each)
    launches_call = prof1["launches"] // 2
    launches_step = launches_call / G

    extras = world == 1 and not args.no_extras
    latency = latency_tl = latency_hr = None

    def tester_latency(one, n_img=8):
        """The reference tester's calling pattern: one image per call, one engine instance, synchronised windows."""
        rl, dl, _ = make_inputs(n_img, H, W, cid + 100)
        for i in range(2):  # size the workspace for B=1; the second call captures the HIP graphs
            o = one.compress(rl[:1], dl[:1])
            one.decompress(o["r_strings"], o["d_strings"], o["shape"])
        enc = dec = 0.0
        nbytes = 0
        for i in range(n_img):
            torch.cuda.synchronize()
            t0 = time.time()
            o = one.compress(rl[i:i + 1], dl[i:i + 1])
            torch.cuda.synchronize()
            t1 = time.time()
            one.decompress(o["r_strings"], o["d_strings"], o["shape"])
            torch.cuda.synchronize()
            t2 = time.time()
            enc += t1 - t0
            dec += t2 - t1
            nbytes += sum(len(s) for k in ("r_strings", "d_strings") for lst in o[k] for s in lst)
        return {"value": round(n_img * H * W / (enc + dec) / 1e6, 4), "unit": "Mpx/s", "images": n_img, "batch": 1,
                "engine_instances": 1, "enc_ms_per_image": round(enc / n_img * 1e3, 2),
                "dec_ms_per_image": round(dec / n_img * 1e3, 2), "bpp_rgb_plus_depth": round(nbytes * 8.0 / (n_img * H * W), 3),
                "definition": "sum(H*W) / (sum(enc) + sum(dec)), torch.cuda.synchronize() around compress() and "
                              "decompress() as in testing/tester_united.py:142-147,180-186"}

    def entropy_stage(one):
        """SURVEY 8(d), second regime: the entropy stage is not MFMA work.  Timed live with HIP events on the stream the kernels
        are launched on, through the C ABI's device-resident entry points, on the model's OWN symbols of one 480x640 pair:
        the two rANS kernels are one serial chain per stream (one wavefront each: ns per symbol is their figure of merit, not
        bytes/s), the checkerboard quantise / index pass is HBM-bound (algorithmic bytes: y, mean, scale in, symbol, index,
        y_hat out = 24 B per coded position, + 4 B per position of the other half the anchor pass zeroes)."""
        import ctypes

        import numpy as np

        from rgbd_amd import ans
        from rgbd_amd._lib import check, lib
        from rgbd_amd.entropy_models import GaussianConditional, get_scale_table

        L = lib()

        def vp(t):
            return ctypes.c_void_p(t.data_ptr())

        rl, dl, _ = make_inputs(1, H, W, cid + 100)
        one.compress(rl, dl)
        pairs = [one.debug_symbols(m_) for m_ in (0, 1)]
        T = int(pairs[0][0].size)
        gcm = GaussianConditional()
        gcm.update_scale_table(get_scale_table(), force=True)
        cdf, sizes, offsets = gcm.numpy_tables()
        tb = ans.Tables(cdf, sizes, offsets)
        st = torch.cuda.Stream(device=dev)
        sym = torch.from_numpy(np.concatenate([pairs[0][0], pairs[1][0], [0]]).astype(np.int32)).to(dev)
        idx = torch.from_numpy(np.concatenate([pairs[0][1], pairs[1][1], [0]]).astype(np.int32)).to(dev)
        base = torch.tensor([0, T], dtype=torch.int64, device=dev)
        cnt = torch.tensor([T, T], dtype=torch.int64, device=dev)
        cap = int(L.rgbd_rans_max_bytes(T)) // 4
        out = torch.zeros(2 * cap, dtype=torch.int32, device=dev)
        words = torch.zeros(2, dtype=torch.int64, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()

        def timed_launch(fn, reps=3):
            best = None
            with torch.cuda.stream(st):
                for _ in range(reps):
                    e0.record(st)
                    fn()
                    e1.record(st)
                    st.synchronize()
                    ms = e0.elapsed_time(e1)
                    best = ms if best is None else min(best, ms)
            return best

        enc_ms = timed_launch(lambda: check(L.rgbd_rans_encode_batch_dev(tb.handle, vp(sym), vp(idx), vp(base), vp(cnt), 2, vp(out), cap,
                                                                         vp(words), vp(err), st.cuda_stream), "encode_batch_dev"))
        assert int(err.item()) == 0
        wn = words.cpu().numpy()
        offs = torch.tensor([cap - int(wn[0]), 2 * cap - int(wn[1])], dtype=torch.int64, device=dev)
        lens = torch.tensor([int(wn[0]), int(wn[1])], dtype=torch.int64, device=dev)
        state = torch.zeros(4, dtype=torch.int64, device=dev)
        got = torch.zeros_like(sym)
        dec_ms = timed_launch(lambda: check(L.rgbd_rans_decode_batch_dev(tb.handle, vp(out), vp(offs), vp(lens), 2, vp(state), 1, vp(idx),
                                                                         vp(got), vp(base), 0, T, st.cuda_stream), "decode_batch_dev"))
        assert torch.equal(got[:2 * T], sym[:2 * T]), "device-batched decode did not return the encoder's symbols"
        # checkerboard quantise / index pass on a 16-image latent batch (what one engine call of the timed region codes per slice)
        n_, c_, h_, w_ = B * G, 192, padded[0] // 16, padded[1] // 16
        gq = torch.Generator(device="cpu").manual_seed(3)
        y_ = (torch.randn(n_, c_, h_, w_, generator=gq) * 6).to(dev)
        mu = torch.randn(n_, c_, h_, w_, generator=gq).to(dev)
        sc = torch.exp(torch.randn(n_, c_, h_, w_, generator=gq)).to(dev)
        m_ = n_ * c_ * h_ * (w_ // 2)
        qs, qi = torch.zeros(m_, dtype=torch.int32, device=dev), torch.zeros(m_, dtype=torch.int32, device=dev)
        yh = torch.zeros_like(y_)
        tabf = np.ascontiguousarray(get_scale_table().numpy(), np.float32)
        f32p = ctypes.POINTER(ctypes.c_float)
        q_ms = timed_launch(lambda: check(L.rgbd_ckbd_quant_index(vp(y_), vp(mu), vp(sc), n_, c_, h_, w_, 1, tabf.ctypes.data_as(f32p),
                                                                  vp(qs), vp(qi), vp(yh), st.cuda_stream), "ckbd_quant_index"), reps=5)
        q_bytes = m_ * 28.0
        return {"bound": "latency (rANS: one wavefront per stream) / hbm (checkerboard pass)",
                "rans_encode_ns_per_symbol": round(enc_ms * 1e6 / T, 1), "rans_decode_ns_per_symbol": round(dec_ms * 1e6 / T, 1),
                "symbols_per_stream": T, "streams": 2, "bytes_per_symbol": round(float(wn.sum()) * 4 / (2 * T), 3),
                "ckbd_quant_index": {"achieved": round(q_bytes / (q_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                     "frac": round(q_bytes / (q_ms * 1e-3) / 8e12, 4), "launch_us": round(q_ms * 1e3, 1),
                                     "algorithmic_bytes": int(q_bytes), "positions": m_},
                "note": "HIP events on the launch stream, C-ABI device entry points (rgbd_rans_*_batch_dev, "
                        "rgbd_ckbd_quant_index), the model's own symbols of one 480x640 pair (stress recipe); separate pass, "
                        "nothing else on the chip"}

    entropy = None
    if extras:
        solo.set_tile_mode("latency")
        entropy = entropy_stage(solo)
        latency = tester_latency(solo)
        latency["weights"] = "synthetic seed 0 (stress recipe: ~22 bpp, wide CDF rows, 17 % escapes)"
        solo.set_tile_mode(tile_mode)
        if model == "ELIC_united":
            # the same pass at a trained model's operating point (rates of ~1.5 bpp per modality, narrow CDF rows): the serial
            # entropy coder, which bounds a single image's latency, is much cheaper per symbol there
            tl = rgbd_amd.modelZoo[model](config=rgbd_amd.model_config(), channel=4).eval()
            tl.load_state_dict(synth.synthetic_state_dict(0, model=model, recipe="trained_like"))
            tl.update(force=True)
            tl = tl.to(dev)
            tl.per_image_streams = True
            latency_tl = tester_latency(tl)
            latency_tl["weights"] = "synthetic seed 0 (trained_like recipe)"
            del tl
            # ... and where a high-quality checkpoint works: scales of 10 ... 100, i.e. scale-table rows of 300 ... 3000 entries
            # (sigma-index 40 ... 57 of 64), which the decoder searches in two hops (coarse first level + one probe)
            hr = rgbd_amd.modelZoo[model](config=rgbd_amd.model_config(), channel=4).eval()
            hr.load_state_dict(synth.synthetic_state_dict(0, model=model, recipe="high_rate"))
            hr.update(force=True)
            hr = hr.to(dev)
            hr.per_image_streams = True
            latency_hr = tester_latency(hr, n_img=4)
            latency_hr["weights"] = "synthetic seed 0 (high_rate recipe: 98 % of the symbols on CDF rows of 300 ... 3000 entries)"
            del hr

    second = None
    others = []
    pairs_in_flight = min(args.workers, calls) * B * G
    sd_main = sd
    net.close()  # (also hands the device its default wait policy back; the STF pool below sets its own)
    if extras and args.workload == "c3_4x480x640" and not os.environ.get("RGBD_BENCH_NO_C5"):
        # BASELINE config 5 (STF_united, Swin transforms) in the same line: one pair per step as the config says -- there a
        # step is the serial coder chain of one image's two streams -- and four pairs per step; each with the roofline of its
        # conv / linear launches (isolated pass, like the headline's) and the CPU oracle beside it.  Each is measured by a
        # child process running this script on that workload (own HIP runtime: the number of hardware queues is fixed when
        # the runtime starts, and 16 instances of this model want fewer than the headline's 20 -- 13.1 vs 10.3 Mpx/s).
        del solo
        torch.cuda.empty_cache()
        # BASELINE config 2 (8 x 256x256 per step) rides along the same way: its own process, its own instance count
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES",) or _USER_QUEUES}
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", "c2_8x256x256", "--steps", str(args.steps), "--warmup",
                   str(args.warmup), "--no-extras", "--no-cpu-baseline"]
            cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            if cp.returncode != 0 or not line:
                raise RuntimeError(f"exit {cp.returncode}: {cp.stderr[-500:]}")
            c = json.loads(line[-1])
            second = {"workload": "c2_8x256x256", "value": c["value"], "unit": "Mpx/s", "ms_per_step": c["ms_per_step"],
                      "images_per_gpu": c["config"]["images_per_gpu"], "image": c["config"]["image"],
                      "engine_instances": c["config"]["engine_instances"], "job_level_frac": c["roofline"]["frac"]}
        except Exception as e:
            print(f"[bench] secondary workload c2_8x256x256 failed: {e}", file=sys.stderr)
            second = {"workload": "c2_8x256x256", "error": str(e)[:300]}
        cpu5 = None
        for name5 in ("c5_stf_1x512x512", "c5_stf_4x512x512"):
            env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES",) or _USER_QUEUES}
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", name5, "--steps", "16", "--warmup", "16", "--workers", "16",
                   "--no-extras", "--cpu-budget", "10"] + (["--no-cpu-baseline"] if (args.no_cpu_baseline or cpu5 is not None) else [])
            try:
                cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                if cp.returncode != 0 or not line:
                    raise RuntimeError(f"exit {cp.returncode}: {cp.stderr[-500:]}")
                c = json.loads(line[-1])
            except Exception as e:  # the headline line must not depend on a secondary workload's child process
                print(f"[bench] secondary workload {name5} failed: {e}", file=sys.stderr)
                others.append({"workload": name5, "error": str(e)[:300]})
                continue
            iso = c["roofline"]["isolated"]
            w = {"workload": name5, "value": c["value"], "unit": "Mpx/s", "steps": c["steps"], "ms_per_step": c["ms_per_step"],
                 "images_per_gpu": c["config"]["images_per_gpu"], "image": c["config"]["image"],
                 "engine_instances": c["config"]["engine_instances"], "codec": c["config"]["codec"],
                 "hbm_workspace_gib": c["config"]["hbm_workspace_gib"],
                 "roofline": {"bound": "mfma", "kernel": "conv_mfma_kernel (conv / deconv / Linear-as-1x1 launches)",
                              "achieved": iso["achieved"], "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": iso["frac"],
                              "traffic": None, "conv_ms_per_step": iso["conv_ms_per_step"],
                              "gflop_per_step": c["roofline"]["gflop_per_step"],
                              "job_level_frac": c["roofline"]["frac"],
                              "definition": "isolated: HIP events around every conv launch of one engine instance alone"}}
            if "cpu_baseline" in c:
                cpu5 = c["cpu_baseline"]
            if cpu5 is not None:
                w["cpu_baseline"] = cpu5
                w["vs_cpu"] = round(w["value"] / cpu5["value"], 2)
            others.append(w)

    traffic = None
    for rnd in ("r05",):  # (the call shape of rounds 1-4 was one step per call: their per-launch bytes do not apply)  # HBM bytes per conv launch from the PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE)
        try:
            name = {"c2_8x256x256": f"{rnd}_pmc_traffic.json", "c3_4x480x640": f"{rnd}_c3_pmc_traffic.json"}[args.workload]
            with open(os.path.join(ROOT, "profiles", name)) as f:
                traffic = round(json.load(f)["hbm_bytes_per_launch"])
            break
        except (OSError, KeyError, ValueError):
            pass
    if rank == 0:
        px = world * B * H * W * args.steps
        bytes_y = sum(len(s) for s in last["r_strings"][0] + last["d_strings"][0])
        job_tflops = flops_step * args.steps / elapsed / 1e12
        iso_tflops = prof_lat["flops"] / (prof_lat["conv_ms"] / 1e3) / 1e12
        iso_tp_tflops = prof1["flops"] / (prof1["conv_ms"] / 1e3) / 1e12
        res = {
            "metric": "RGB-D Mpixels/s encode+decode",
            "value": round(px / elapsed / 1e6, 4),
            "unit": "Mpx/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload,
                       "codec": "ELIC_united ch4 q=2_2 (N=192,M=320)" if model == "ELIC_united" else "STF_united ch4 (N=192,M=384)",
                       "images_per_gpu": B, "steps_per_call": G, "images_per_call": B * G, "calls": calls,
                       "image": [H, W], "padded": list(padded), "weights": f"synthetic seed 0 ({recipe} recipe)",
                       "streams": "per image", "y_bytes_last_batch": bytes_y, "engine_instances": args.workers,
                       "conv_tiles": tile_mode, "host_cores_busy_per_rank": host_cores,
                       "warmup_steps_run": host.get("warmup_steps_run", 0),
                       "launch": "HIP graph per call shape" if not os.environ.get("RGBD_NO_GRAPH") else "eager",
                       # what the throughput costs: image pairs held at once and HBM workspace of all engine instances
                       # of this rank (the 0.6 GB of packed weights are shared and not included)
                       "pairs_in_flight": pairs_in_flight, "hbm_workspace_gib": workspace_gib,
                       "hbm_workspace_gib_per_instance": round(workspace_gib / max(args.workers, 1), 3)},
            "parity": parity_state(),
            # `achieved`: algorithmic conv FLOPs of the timed steps / wall time of the timed region (job level, a lower
            # bound on MFMA utilisation: the wall clock also holds every other kernel).  With several engine instances
            # sharing the chip a per-launch event bracket would also contain CU time-sharing, so the per-launch figure is
            # measured for one instance alone in a separate pass (`isolated`; agrees with `rocprofv3 --stats` of
            # `--workers 1`, committed under profiles/).
            "roofline": {"bound": "mfma", "kernel": "conv_mfma_kernel (all conv/deconv layers)",
                         "achieved": round(job_tflops, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(job_tflops / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "traffic_note": "HBM bytes per conv launch, rocprofv3 PMC passes committed under profiles/ "
                                         "(not collectable from inside this process)",
                         "definition": "algorithmic conv FLOPs of the timed steps / wall time of the timed region, per GPU",
                         # FLOPs the launches execute / algorithmic FLOPs: < 1 because a checkerboard-output launch computes one
                         # half of its layer and an anchor-input launch half of the taps as well (what the MFMA-busy counter
                         # sees is `frac` x this)
                         "executed_over_algorithmic": round(prof1["flops_executed"] / max(prof1["flops"], 1.0), 4),
                         "executed_frac": round(job_tflops / PEAK_FP32_MFMA_TFLOPS * prof1["flops_executed"] / max(prof1["flops"], 1.0), 4),
                         "hbm": None if traffic is None else {
                             "achieved": round(traffic * launches_call / (elapsed / calls) / 1e9, 1), "peak": 8000.0,
                             "unit": "GB/s", "frac": round(traffic * launches_call / (elapsed / calls) / 8e12, 4),
                             "note": "PMC HBM bytes of the conv launches of one step / step time: the path is MFMA-bound"},
                         "launches_per_call": launches_call, "launches_per_step": round(launches_step, 2),
                         "gflop_per_step": round(flops_step / 1e9, 2),
                         "isolated": {"achieved": round(iso_tflops, 3), "frac": round(iso_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
                                      "avg_launch_us": round(prof_lat["conv_ms"] * 1e3 / max(prof_lat["launches"], 1), 2),
                                      "executed_frac": round(prof_lat["flops_executed"] / (prof_lat["conv_ms"] / 1e3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                      "conv_ms_per_step": round(prof_lat["conv_ms"] / 2 / G, 3), "conv_tiles": "latency",
                                      "note": "HIP events around every conv launch on its stream, single engine instance, "
                                              "no concurrent kernels, separate pass after the timed region, latency tiles "
                                              "(what a lone instance runs); profiles/r05_bench_w1_summary.txt"},
                         "isolated_timed_tiles": {"achieved": round(iso_tp_tflops, 3),
                                                  "frac": round(iso_tp_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
                                                  "conv_ms_per_step": round(prof1["conv_ms"] / 2 / G, 3), "conv_tiles": tile_mode,
                                                  "note": "the same pass with the tiles the timed region ran"}},
        }
        if entropy is not None:
            res["roofline"]["entropy"] = entropy
        if sustained is not None:
            res["sustained"] = sustained
        if latency is not None:
            res["latency"] = latency
        if latency_tl is not None:
            res["latency_trained_like"] = latency_tl
        if latency_hr is not None:
            res["latency_high_rate"] = latency_hr
        if second is not None:
            res["workloads"] = [{"workload": args.workload, "value": res["value"], "unit": "Mpx/s",
                                 "ms_per_step": res["ms_per_step"], "images_per_gpu": B, "image": [H, W]}, second] + others
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(sd_main, H, W, cid, model, seconds_budget=args.cpu_budget, batch8=args.cpu_budget >= 20.0)
            res["cpu_baseline"] = cpu
            res["vs_cpu"] = {"throughput": round(res["value"] / cpu["value"], 2),
                             "latency_tester_semantics": None if latency is None else round(latency["value"] / cpu["value"], 2),
                             "latency_tester_semantics_trained_like": None if latency_tl is None else round(latency_tl["value"] / cpu["value"], 2),
                             "note": "all over the best CPU-oracle leg (cpu_baseline.legs: thread counts x batch sizes).  The "
                                     "north-star target (>= 40x the CPU reference) is met as BATCH THROUGHPUT (`throughput`: "
                                     "engine_instances batches in flight).  The reference's own calling pattern -- one image "
                                     "per call -- is bounded by the stream format, not by this implementation's kernels: one "
                                     "rANS state per modality, and Bi-CEE (models/elic_united.py:454-541) decodes the two "
                                     "modalities' 20 parts strictly one after the other, ~0.8 M serial symbol steps per "
                                     "480x640 pair"}
            if second is not None and "value" in second:
                cpu2 = cpu_baseline(sd_main, 256, 256, 2, model, seconds_budget=8.0, batch8=False)
                second["cpu_baseline"] = cpu2
                second["vs_cpu"] = round(second["value"] / cpu2["value"], 2)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
