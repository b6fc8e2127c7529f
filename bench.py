#!/usr/bin/env python3
"""Headline benchmark: RGB-D Mpixels/s, encode+decode, ELIC_united q=2_2 on MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = compress() + decompress() of one batch of synthetic RGB-D pairs per rank (weak scaling: every rank codes
its own batch; no collective on the data path, only the gather of the finished streams).  Inputs are resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
    "c3_4x480x640": (4, 480, 640, 3, "ELIC_united"),
    "c5_stf_1x512x512": (1, 512, 512, 5, "STF_united"),  # BASELINE config 5 (Swin transforms)
}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table (dense f32 matrix)


def cpu_baseline(sd, H, W, cid, model="ELIC_united", seconds_budget=25.0):
    """The oracle (CPU restatement of the reference path: PyTorch-CPU eager + C coder) timed on this host, B=1, on single
    pairs of the workload's image size (replicate-padded to multiples of 64 like the harness does)."""
    import torch

    from oracle import elic_oracle as eo
    from rgbd_amd import synth

    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    orc = eo.OracleCodec(sd) if model == "ELIC_united" else eo.oracle_stf(sd)
    orc.update()
    done, spent, best = 0, 0.0, None
    while spent < seconds_budget and done < 6:
        r, d = synth.synthetic_batch(1, H, W, config_id=cid, start=done)
        r, d = torch.from_numpy(r), torch.from_numpy(d)
        r, d = eo.pad_replicate0(r), eo.pad_replicate0(d)
        t0 = time.time()
        out = orc.compress(r, d)
        orc.decompress(out["r_strings"], out["d_strings"], out["shape"])
        dt = time.time() - t0
        spent += dt
        done += 1
        best = dt if best is None else min(best, dt)
    return {"value": round(H * W / best / 1e6, 5), "unit": "Mpx/s", "cores": cores, "kind": "port",
            "sample": f"best of {done} single {H}x{W} pairs, enc+dec, B=1 (tester semantics), torch CPU {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="c2_8x256x256", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workers", type=int, default=16, help="engine instances (HIP streams) per GPU; 1 = no overlap")
    ap.add_argument("--tile-mode", default="auto", choices=["auto", "latency", "throughput"],
                    help="conv tile tables: auto = throughput tiles when >= 4 engine instances share the GPU (CodecPool's rule)")
    args = ap.parse_args()
    if not _USER_QUEUES:
        os.environ["GPU_MAX_HW_QUEUES"] = str(max(24, args.workers + 8))

    import torch

    import rgbd_amd
    from rgbd_amd import CodecPool, distributed, synth

    rank, world, local = distributed.init_from_env()
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    B, H, W, cid, model = WORKLOADS[args.workload]
    sd = synth.synthetic_state_dict(0, model=model)
    # per-image stream sets (the unit that shards across GPUs); W engine instances overlap one group's serial coder
    # phases with another group's convolutions
    net = CodecPool(sd, config=rgbd_amd.model_config(), workers=args.workers, device=dev, per_image_streams=True,
                    model_cls=rgbd_amd.modelZoo[model])

    if args.tile_mode != "auto":
        for n_ in net.nets:
            n_.set_tile_mode(args.tile_mode)
    tile_mode = args.tile_mode if args.tile_mode != "auto" else ("throughput" if args.workers >= 4 else "latency")

    r, d = synth.synthetic_batch(B, H, W, config_id=cid, start=rank * B)
    rgb, depth = torch.from_numpy(r).to(dev), torch.from_numpy(d).to(dev)
    ph, pw = (-H) % 64, (-W) % 64
    if ph or pw:  # dataset/utils.py:58-67 "replicate0"
        rgb = torch.nn.functional.pad(rgb, (0, pw, 0, ph), mode="replicate")
        depth = torch.nn.functional.pad(depth, (0, pw, 0, ph), mode="replicate")
    rgb, depth = rgb.contiguous(), depth.contiguous()

    def run(nsteps):
        # every step codes one full batch (compress + decompress); the W engine instances keep W steps in flight, so
        # one step's serial coder phases overlap another step's convolutions.  All nsteps finish before this returns.
        res = net.roundtrip_many([(rgb, depth)] * nsteps)
        if world > 1:  # the job's only exchange: the finished streams of these steps to every rank (one RCCL all_gather)
            distributed.gather_streams([s for o, _, _ in res for s in o["r_strings"][0] + o["d_strings"][0]])
        return res

    if args.warmup:
        run(max(args.warmup, min(args.workers, args.steps)))  # every engine instance sizes its workspace once
    net.set_profile(True)
    distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(args.steps)
    torch.cuda.synchronize()
    distributed.barrier()
    elapsed = distributed.max_over_ranks(time.perf_counter() - t0)
    out = [res[-1][0]]
    prof = net.profile_read()
    net.set_profile(False)
    # the same kernels with nothing else on the chip (one engine instance, two more steps): event brackets in the timed
    # region above also contain the time a conv launch spends sharing CUs with the other instances' kernels
    solo = net.nets[0]
    solo.set_profile(True)
    for _ in range(2):
        o = solo.compress(rgb, depth)
        solo.decompress(o["r_strings"], o["d_strings"], o["shape"])
    prof1 = solo.profile_read()
    solo.set_profile(False)

    traffic = None
    try:  # HBM bytes per conv launch from the PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), see profiles/
        name = {"c2_8x256x256": "r01_pmc_traffic.json", "c3_4x480x640": "r01_c3_pmc_traffic.json"}[args.workload]
        with open(os.path.join(ROOT, "profiles", name)) as f:
            traffic = round(json.load(f)["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    if rank == 0:
        px = world * B * H * W * args.steps
        bytes_y = sum(len(s) for o in out for s in o["r_strings"][0] + o["d_strings"][0])
        conv_s = prof["conv_ms"] / 1e3
        achieved = prof["flops"] / conv_s / 1e12 if conv_s > 0 else 0.0
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
            "config": {"workload": args.workload, "codec": "ELIC_united ch4 q=2_2 (N=192,M=320)" if model == "ELIC_united" else "STF_united ch4 (N=192,M=384)", "images_per_gpu": B,
                       "image": [H, W], "padded": [H + ph, W + pw], "weights": "synthetic seed 0 (stress recipe)",
                       "streams": "per image", "y_bytes_last_batch": bytes_y, "engine_instances": args.workers,
                       "conv_tiles": tile_mode},
            # `achieved`: conv FLOPs of the timed steps / wall time of the timed region -- with several engine instances
            # sharing the chip a per-launch event bracket also contains the time the launch spends sharing CUs with other
            # instances' kernels, so the per-launch figures are reported twice: as measured inside the timed region
            # (`timed_region_brackets`, agrees with `rocprofv3 --stats` of this command) and for one instance alone
            # (`isolated`, agrees with `rocprofv3 --stats` of `--workers 1`).
            "roofline": {"bound": "mfma", "kernel": "conv_mfma_kernel (all conv/deconv layers)",
                         "achieved": round(prof["flops"] / elapsed / 1e12, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(prof["flops"] / elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                         "traffic": traffic,
                         "traffic_note": "HBM bytes per conv launch, rocprofv3 PMC passes committed under profiles/ "
                                         "(not collectable from inside this process)",
                         "definition": "algorithmic conv FLOPs of the timed steps / wall time of the timed region, per GPU "
                                       "(a lower bound on MFMA utilisation: the wall clock also holds every other kernel)",
                         "hbm": None if traffic is None else {
                             "achieved": round(traffic * (prof["launches"] / max(args.steps, 1)) / (elapsed / args.steps) / 1e9, 1),
                             "peak": 8000.0, "unit": "GB/s",
                             "frac": round(traffic * (prof["launches"] / max(args.steps, 1)) / (elapsed / args.steps) / 8e12, 4),
                             "note": "PMC HBM bytes of the conv launches of one step / step time: the path is MFMA-bound, "
                                     "not HBM-bound"},
                         "launches_per_step": prof["launches"] // max(args.steps, 1),
                         "gflop_per_step": round(prof["flops"] / max(args.steps, 1) / 1e9, 2),
                         "timed_region_brackets": {"achieved": round(achieved, 3),
                                                   "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                                                   "avg_launch_us": round(prof["conv_ms"] * 1e3 / max(prof["launches"], 1), 2),
                                                   "conv_ms_per_step": round(prof["conv_ms"] / max(args.steps, 1), 3),
                                                   "note": f"HIP events around every conv launch while {args.workers} engine "
                                                           "instances share the chip (durations include CU time-sharing)"},
                         "isolated": {"achieved": round(prof1["flops"] / (prof1["conv_ms"] / 1e3) / 1e12, 3),
                                      "frac": round(prof1["flops"] / (prof1["conv_ms"] / 1e3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                      "avg_launch_us": round(prof1["conv_ms"] * 1e3 / max(prof1["launches"], 1), 2),
                                      "conv_ms_per_step": round(prof1["conv_ms"] / 2, 3),
                                      "note": "same launches, single engine instance, no concurrent kernels"}},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sd, H, W, cid, model)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
