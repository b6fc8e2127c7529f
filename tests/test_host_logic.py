"""Host-side logic that needs no GPU: C-ABI surface, table construction, checkpoint interface, container / pad / PSNR
arithmetic of the harness (pinned by goldens captured from the reference)."""
import ctypes
import hashlib
import io
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import elic_oracle as eo


def test_library_exports_every_declared_symbol():
    from rgbd_amd import _lib

    L = _lib.lib()  # raises if a declared symbol is missing
    assert L.rgbd_abi_version() == 1
    header = open(os.path.join(ROOT, "include", "rgbd_amd.h")).read()
    declared = set(re.findall(r"\b(rgbd_[a-z0-9_]+)\s*\(", header))
    raw = ctypes.CDLL(_lib._SO)
    for name in sorted(declared):
        getattr(raw, name)
    assert declared == set(_lib.EXPORTS)


def test_abi_argument_errors_without_gpu():
    from rgbd_amd._lib import lib

    L = lib()
    out = (ctypes.c_uint32 * 4)()
    assert L.rgbd_pmf_to_quantized_cdf(None, 3, 16, out) == -22
    h = ctypes.c_void_p()
    bad = (ctypes.c_int32 * 2)(16, 17)
    assert L.rgbd_elic_create(192, 320, bad, 2, ctypes.byref(h)) == -22  # slices must sum to M, multiples of 16
    assert L.rgbd_elic_compress(None, None, None, 1, 64, 64, 1, None) != 0
    assert L.rgbd_rans_max_bytes(100) >= 4 * (100 + 2)


def test_host_cdf_quantiser_matches_reference(kat):
    from rgbd_amd import ans

    for k in range(4):
        assert ans.pmf_to_quantized_cdf(kat[f"pmf{k}"].tolist(), 16) == kat[f"pmf{k}_cdf"].tolist()


def test_update_builds_reference_tables(kat, synth_sd):
    import rgbd_amd

    net = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
    net.load_state_dict(synth_sd)
    assert net.update(force=True) is True
    gc = net.rgb_gaussian_conditional
    assert np.array_equal(gc.quantized_cdf.numpy(), kat["gc_cdf"])
    assert np.array_equal(gc.cdf_length.numpy(), kat["gc_sizes"]) and np.array_equal(gc.offset.numpy(), kat["gc_offsets"])
    for m in ("rgb", "depth"):
        eb = getattr(net, f"{m}_entropy_bottleneck")
        assert np.array_equal(eb.quantized_cdf.numpy(), kat[f"{m}_eb_cdf"])
        assert np.array_equal(eb.cdf_length.numpy(), kat[f"{m}_eb_sizes"])
    assert net.update(force=False) is True  # bottlenecks always rebuild (entropy_models.py:320-325), as in the reference
    # state_dict round trip keeps every reference key, including the table buffers
    sd = net.state_dict()
    assert list(sd.keys()) == list(synth_sd.keys())
    assert sd["rgb_gaussian_conditional._quantized_cdf"].shape == (64, 3133)
    net2 = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4)
    net2.load_state_dict(sd, strict=True)
    assert np.array_equal(net2.rgb_gaussian_conditional.quantized_cdf.numpy(), kat["gc_cdf"])
    assert sum(p.numel() for p in net2.parameters()) == 149532369


def test_checkpoint_interface_errors(synth_sd):
    import rgbd_amd
    from rgbd_amd._lib import RgbdError

    net = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4)
    bad = dict(synth_sd)
    bad.pop("g_a.rgb_analysis_transform.0.weight")
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad, strict=True)
    wrong = dict(synth_sd)
    wrong["g_a.rgb_analysis_transform.0.weight"] = torch.zeros(192, 3, 3, 3)
    with pytest.raises(RuntimeError):
        net.load_state_dict(wrong)
    with pytest.raises(RgbdError):
        net.to("cpu")
    fresh = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4)
    with pytest.raises(ValueError, match="Run update"):
        fresh.rgb_gaussian_conditional.check()
    with pytest.raises(RgbdError):
        fresh.compress(torch.zeros(1, 3, 64, 64), torch.zeros(1, 1, 64, 64))


@pytest.mark.parametrize("name", ["a_128x192", "b_100x150", "d_256x256"])
def test_container_bytes_and_bpp(name, tmp_path):
    from rgbd_amd import ioutils

    g = load_golden(name)
    H, W = int(g["H"]), int(g["W"])
    for k in ("r", "d"):
        strings = [[g[f"{k}_y"].tobytes()], [g[f"{k}_z0"].tobytes()]]
        fn = tmp_path / f"{name}_{k}.bin"
        with open(fn, "wb") as f:
            ioutils.write_uints(f, (H, W))
            ioutils.write_body(f, tuple(g["shape"]), strings)
        data = open(fn, "rb").read()
        assert hashlib.sha256(data).hexdigest()[:16] == g[f"{k}_container_sha"].tobytes().decode()
        assert data == eo.container_bytes(H, W, tuple(g["shape"]), strings)
        assert ioutils.filesize(fn) * 8.0 / (H * W) == g["bpp"][0 if k == "r" else 1]
        with open(fn, "rb") as f:
            assert ioutils.read_uints(f, 2) == (H, W)
            back, shape = ioutils.read_body(f)
        assert back == strings and tuple(shape) == tuple(g["shape"])


def test_pad_crop_like_reference():
    from rgbd_amd import datautils

    x = torch.arange(2 * 3 * 100 * 150, dtype=torch.float32).reshape(2, 3, 100, 150)
    p = datautils.pad(x, "replicate0")
    assert tuple(p.shape[-2:]) == tuple(load_golden("b_100x150")["padded"]) == (128, 192)
    assert torch.equal(p, eo.pad_replicate0(x))
    assert torch.equal(p[:, :, 100:, :150], x[:, :, 99:100, :].expand(-1, -1, 28, -1))
    assert torch.equal(datautils.crop(p, "replicate0", (100, 150)), x)
    assert datautils.pad(torch.zeros(1, 1, 128, 192), "replicate0").shape[-2:] == (128, 192)
    c = datautils.pad(x, "reflect1")
    assert torch.equal(datautils.crop(c, "reflect1", (100, 150)), x)


def test_psnr_matches_reference_arithmetic():
    from rgbd_amd import metrics

    g = load_golden("a_128x192")
    from rgbd_amd import synth

    r, d = synth.synthetic_batch(1, 128, 192, config_id=9)
    assert abs(metrics.psnr(torch.from_numpy(g["xhat_r"]), torch.from_numpy(r)) - g["psnr"][0]) < 1e-9
    assert abs(metrics.psnr(torch.from_numpy(g["xhat_d"]), torch.from_numpy(d)) - g["psnr"][1]) < 1e-9
    x = torch.rand(1, 3, 192, 192)
    assert abs(metrics.ms_ssim(x, x) - 1.0) < 1e-6


def test_synthetic_weights_are_reproducible():
    from rgbd_amd import synth

    a = synth.uniform_like("rgb", 9000, (4,), 0.0, 1.0)
    assert a.dtype == np.float32 and np.allclose(a, synth.uniform_like("rgb", 9000, (4,), 0.0, 1.0))
    w = synth.make_tensor("g_a.rgb_analysis_transform.0.weight", synth.elic_united_entries()["g_a.rgb_analysis_transform.0.weight"], 0)
    assert hashlib.sha256(w.tobytes()).hexdigest()[:16] == hashlib.sha256(
        synth.synthetic_state_dict.__globals__["make_tensor"]("g_a.rgb_analysis_transform.0.weight",
                                                              synth.elic_united_entries()["g_a.rgb_analysis_transform.0.weight"], 0).tobytes()).hexdigest()[:16]


@pytest.mark.parametrize("table", ["tile_table.h", "tile_table_loaded.h"])
def test_tile_table_is_well_formed(table):
    """csrc/tile_table*.h are generated by tools/tune_tiles.py: 14 integers per entry, tile shapes that exist, unique keys."""
    import os
    import re

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "learning-based-rgb-d-image-compression_amd", "csrc", table)
    tiles = ({(2, m, 8) for m in (3, 2, 1)} | {(2, m, n) for n in (4, 2, 1) for m in (5, 4, 3, 2, 1)} |
             {(1, m, n) for n in (4, 2, 1) for m in (3, 2, 1)})
    keys = set()
    for ln in open(path):
        ln = ln.strip()
        if not ln.startswith("{"):
            continue
        v = [int(x) for x in re.findall(r"-?\d+", ln)]
        assert len(v) == 14, ln
        key, (wm, mt, nt, kc, dma) = tuple(v[:9]), v[9:]
        assert key not in keys, ln
        keys.add(key)
        # staging modes: 0 registers, 1 / 2 / 3 direct-to-LDS double buffer (78 / 52 / 38 KiB cap), 4 / 5 ring of four / three
        # direct-to-LDS stages (single-tap layers; needs whole waves of patch slots: 64 | pixels per tile)
        assert (wm, mt, nt) in tiles and kc in (16, 64) and dma in (0, 1, 2, 3, 4, 5) and not (dma and kc == 64), ln
        if dma >= 4:
            assert key[5] == 1 and key[6] == 1 and key[7] == 1 and (16 * nt * (2 if wm == 2 else 4)) % 64 == 0, ln
        assert all(x > 0 for x in key) and key[3] % 16 == 0 and key[4] % 16 == 0 and key[7] in (1, 4, 11, 21), ln  # 11 / 21: checkerboard-output launches (nphase + 10 * ckbd)
    assert keys


def test_ms_ssim_against_the_definition():
    """utils/metrics.py:13 calls pytorch_msssim (1.0.0, not installable here): the product's torch restatement is checked
    against an independent numpy fp64 statement of the published definition (oracle/msssim_ref.py) -- unpinned against the
    package itself, which the harness log line says too."""
    import torch

    from oracle import msssim_ref
    from rgbd_amd import metrics, synth

    for (h, w, cid, noise) in ((176, 208, 3, 0.05), (161, 193, 4, 0.2), (256, 256, 5, 0.01)):
        r, d = synth.synthetic_pair(cid, h, w, config_id=cid, smooth=True)
        a = torch.from_numpy(r)[None].float()
        rng = np.random.RandomState(cid)
        b = (a + noise * torch.from_numpy(rng.standard_normal(a.shape).astype(np.float32))).clamp(0, 1)
        got = metrics.ms_ssim(a, b, data_range=1.0)
        want = msssim_ref.ms_ssim(a.numpy(), b.numpy(), 1.0)
        assert abs(got - want) < 2e-5, (h, w, got, want)
        assert 0.0 < got < 1.0
        a1 = torch.from_numpy(d)[None].float()
        assert abs(metrics.ms_ssim(a1, a1) - 1.0) < 1e-6
    with pytest.raises(ValueError):
        metrics.ms_ssim(torch.zeros(1, 1, 160, 300), torch.zeros(1, 1, 160, 300))


def test_load_state_dict_diagnostics(caplog):
    """models/elic_united.py:613-620 tries strict loading first and prints what did not match before falling back; here a
    partial checkpoint is accepted but logged, a DDP 'module.' prefix is stripped, and a checkpoint that matches no
    parameter at all is an error instead of a silent run on the synthetic initialisation."""
    import logging

    import rgbd_amd
    from rgbd_amd import synth

    sd = synth.synthetic_state_dict(0)
    m = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
    m.load_state_dict(sd, strict=True)  # the full reference key set loads strictly
    # a DataParallel / DDP checkpoint
    m2 = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
    m2.load_state_dict({"module." + k: v for k, v in sd.items()}, strict=True)
    k0 = "g_a.rgb_analysis_transform.0.weight"
    assert torch.equal(m2.state_dict()[k0], sd[k0])
    # partial checkpoint: accepted, but not silently
    part = {k: v for k, v in sd.items() if not k.startswith("h_s.")}
    part["not.a.key"] = torch.zeros(1)
    with caplog.at_level(logging.WARNING, logger="rgbd_amd"):
        m3 = rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).eval()
        m3.load_state_dict(part)
    text = caplog.text
    assert "missing keys" in text and "h_s." in text and "unexpected keys" in text and "not.a.key" in text
    with pytest.raises(RuntimeError):
        m3.load_state_dict(part, strict=True)
    # nothing matches: an error, not a model that silently runs on its own initialisation
    with pytest.raises(RuntimeError, match="none of the"):
        rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).load_state_dict({"encoder.w": torch.zeros(3)})
    # wrong shape
    bad = dict(sd)
    bad[k0] = torch.zeros(5, 5)
    with pytest.raises(RuntimeError, match="size mismatch"):
        rgbd_amd.ELIC_united(config=rgbd_amd.model_config(), channel=4).load_state_dict(bad)


def test_image_read_semantics_like_the_reference_dataset(tmp_path):
    """dataset/testDataset.py:36-61 (cv2.imread UNCHANGED + BGR->RGB, /255; depth divided by 10000 / 100000 / 255 by its
    maximum, with the reference's strict inequalities) restated over PIL: PNG decoding is lossless, so the tensors are the
    reference's for 8-bit RGB and 8/16-bit depth files."""
    from PIL import Image

    from rgbd_amd import tester

    rng = np.random.RandomState(3)
    rgb = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    Image.fromarray(rgb).save(tmp_path / "c.png")
    t = tester.load_image(tmp_path / "c.png", "RGB")
    assert t.shape == (3, 37, 53) and t.dtype == torch.float32
    assert torch.equal(t, torch.from_numpy(rgb.transpose(2, 0, 1)).float() / 255.0)  # R, G, B order, exact
    for mx, div in ((200, 255.0), (255, 255.0), (256, 10000.0), (9999, 10000.0), (10000, 255.0), (10001, 100000.0),
                    (65535, 100000.0)):
        d = rng.randint(0, mx + 1, (21, 34)).astype(np.uint16)
        d[0, 0] = mx
        Image.fromarray(d).save(tmp_path / "d.png")
        td = tester.load_image(tmp_path / "d.png", "L")
        assert td.shape == (1, 21, 34)
        assert torch.equal(td, torch.from_numpy(d.astype("float32"))[None] / div), mx
    # the harness writes depth reconstructions as 16-bit PNGs (tester_united.py:101-109)
    x = torch.rand(1, 1, 8, 9)
    tester.save_depth16(x, tmp_path / "r.png", 10000)
    back = np.array(Image.open(tmp_path / "r.png"))
    assert back.dtype == np.uint16 and np.array_equal(back, (x * 10000).squeeze().numpy().astype("uint16"))


def test_trained_like_recipe_is_defined_and_reproducible():
    """The second synthetic weight recipe (bench.py `latency_trained_like`): same generator, other gains; ELIC_united only."""
    from rgbd_amd import synth

    a = synth.synthetic_state_dict(0, recipe="trained_like", as_torch=False)
    b = synth.synthetic_state_dict(0, recipe="trained_like", as_torch=False)
    s = synth.synthetic_state_dict(0, as_torch=False)  # stress recipe (default)
    k = "g_a.rgb_analysis_transform.16.weight"
    assert np.array_equal(a[k], b[k]) and not np.array_equal(a[k], s[k])
    assert np.allclose(a[k] * 8.0, s[k])  # gains 6 vs 48 on the last analysis conv
    kb = "rgb_entropy_parameters_anchor.0.fusion.4.bias"
    assert np.allclose(a[kb][:16], 0.25) and np.allclose(s[kb][:16], 1.0)
    untouched = "g_s.rgb_synthesis_transform.3.branch.2.weight"
    assert np.array_equal(a[untouched], s[untouched])
    with pytest.raises(ValueError):
        synth.synthetic_state_dict(0, recipe="trained_like", model="STF_united")
    with pytest.raises(ValueError):
        synth.synthetic_state_dict(0, recipe="nope")


def test_high_rate_recipe_is_defined_and_reproducible():
    """The third synthetic weight recipe (bench.py `latency_high_rate`: symbols on CDF rows of 300 ... 3000 entries)."""
    from rgbd_amd import synth

    a = synth.synthetic_state_dict(0, recipe="high_rate", as_torch=False)
    b = synth.synthetic_state_dict(0, recipe="high_rate", as_torch=False)
    p = synth.synthetic_state_dict(0, recipe="plain", as_torch=False)
    k = "g_a.depth_analysis_transform.16.weight"
    assert np.array_equal(a[k], b[k]) and np.allclose(a[k], p[k] * synth.HIGH_RATE_GAINS[0])
    kb = "depth_entropy_parameters_nonanchor.4.fusion.4.bias"
    assert np.allclose(a[kb][:192], synth.HIGH_RATE_GAINS[4]) and np.array_equal(a[kb][192:], p[kb][192:])
    with pytest.raises(ValueError):
        synth.synthetic_state_dict(0, recipe="high_rate", model="ELIC_united_R2D")


def test_balanced_workers_fill_every_round():
    """Engine instances for a job of K batches: as few rounds as the cap allows, every round full."""
    from rgbd_amd.pool import balanced_workers

    assert balanced_workers(20) == 20 and balanced_workers(48) == 16 and balanced_workers(24) == 12
    assert balanced_workers(21) == 11 and balanced_workers(1) == 1 and balanced_workers(0) == 1
    assert balanced_workers(20, max_workers=16) == 10 and balanced_workers(7, max_workers=4) == 4
    for k in range(1, 200):
        w = balanced_workers(k)
        rounds = -(-k // w)
        assert 1 <= w <= 20 and rounds == -(-k // 20) and rounds * w - k < rounds  # no round short by a whole instance


def test_codec_pool_refuses_more_instances_than_streams():
    """torch's side-stream pool holds 32 streams per device; a pool that aliased two instances onto one stream would
    corrupt graph captures, so it is refused before anything touches the GPU."""
    import rgbd_amd
    from rgbd_amd.pool import CodecPool

    with pytest.raises(ValueError, match="at most 32"):
        CodecPool({}, config=rgbd_amd.model_config(), workers=33, device="cuda:0")


def test_bench_host_cpu_info_and_worker_rule(tmp_path, monkeypatch):
    """bench.py's host description (what cpu_baseline reports: model, usable logical / physical CPUs, cgroup quota) and the
    engine-instance rule it shares with CodecPool through the torch-free sched module."""
    import importlib.util

    from conftest import ROOT

    spec = importlib.util.spec_from_file_location("_bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    info = bench.host_cpu_info()
    assert info["logical_allowed"] >= 1 and 1 <= info["physical_allowed"] <= info["logical_allowed"]
    assert isinstance(info["model"], str) and (info["cgroup_cpu_quota"] is None or info["cgroup_cpu_quota"] > 0)
    from rgbd_amd.pool import balanced_workers
    from rgbd_amd.sched import balanced_workers as bw2

    assert balanced_workers is bw2
    assert [bw2(n) for n in (1, 5, 20, 21, 40, 48, 64)] == [1, 5, 20, 11, 20, 16, 16]


def test_hw_queue_count_outside_the_tested_range_is_refused():
    """64 hardware queues made launches on other streams fail; the pool / harness refuse it up front."""
    from rgbd_amd.sched import MAX_SAFE_HW_QUEUES, check_hw_queues

    assert check_hw_queues({}) == 0
    assert check_hw_queues({"GPU_MAX_HW_QUEUES": "40"}) == 40
    assert check_hw_queues({"GPU_MAX_HW_QUEUES": str(MAX_SAFE_HW_QUEUES)}) == MAX_SAFE_HW_QUEUES
    with pytest.raises(ValueError, match="GPU_MAX_HW_QUEUES=64"):
        check_hw_queues({"GPU_MAX_HW_QUEUES": "64"})
    with pytest.raises(ValueError, match="not an integer"):
        check_hw_queues({"GPU_MAX_HW_QUEUES": "many"})


def test_tile_override_csv_is_parsed_on_the_host():
    """rgbd_debug_tile_override (tools/tune_insitu.py): 14 integers per line, "" clears; a short line is refused.  Pure host code."""
    from rgbd_amd._lib import lib

    L = lib()
    assert L.rgbd_debug_tile_override(b"16,32,40,192,384,25,1,101,1,2,2,8,16,1\n32,256,320,192,192,9,1,101,1,1,3,4,16,1") == 0
    assert L.rgbd_debug_tile_override(b"16,32,40,192,384,25,1,101,1,2,2,8") == -22
    assert L.rgbd_debug_tile_override(b"") == 0
