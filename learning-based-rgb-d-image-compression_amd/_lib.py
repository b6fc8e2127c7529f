"""Loader (and builder) of librgbd_amd.so, the C-ABI HIP library of this package (include/rgbd_amd.h).

There is no CPU fallback: if the library is missing or a HIP call fails, the operators raise.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librgbd_amd.so")
_SO = os.environ.get("RGBD_AMD_LIB", _SO)  # A/B builds: point at another librgbd_amd.so
_SRCS = ["conv_mfma.hip", "conv_mfma_blk.hip", "pointwise.hip", "swin.hip", "entropy.hip", "metrics.hip", "coder_abi.hip", "engine.hip", "engine_abi.hip"]
_LIB = None

ERRORS = {-22: "invalid argument", -12: "out of memory", -5: "HIP runtime error", -28: "buffer too small",
          -1: "wrong call order / missing weights or tables"}


class RgbdError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        raise RgbdError(f"librgbd_amd: {what} failed with {rc} ({ERRORS.get(rc, 'unknown')})")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into librgbd_amd.so (in-tree, next to this file): one object per source
    (only stale ones are recompiled, up to four at a time), then one link."""
    from concurrent.futures import ThreadPoolExecutor

    csrc = os.path.join(_HERE, "csrc")
    objdir = os.path.join(csrc, "build")
    hdrs = [os.path.join(csrc, "common.h"), os.path.join(csrc, "exact_math.h"),
            os.path.join(os.path.dirname(_HERE), "include", "rgbd_amd.h")]
    body = os.path.join(csrc, "conv_mfma_body.h")
    extra = {"conv_mfma.hip": [body, os.path.join(csrc, "tile_table.h"), os.path.join(csrc, "tile_table_loaded.h"),
                               os.path.join(csrc, "tile_table_blk.h"), os.path.join(csrc, "tile_table_blk_loaded.h")],
             "conv_mfma_blk.hip": [body]}
    hdrs.append(os.path.join(csrc, "splitk_table.h"))
    extra["coder_abi.hip"] = [os.path.join(csrc, "engine_internal.h")]
    extra["engine.hip"] = extra["engine_abi.hip"] = [os.path.join(csrc, "engine_internal.h"), os.path.join(csrc, "engine.h")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for name in _SRCS:
        src, obj = os.path.join(csrc, name), os.path.join(objdir, name + ".o")
        objs.append(obj)
        deps = [src] + hdrs + extra.get(name, [])
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in deps):
            # (conv_mfma_blk.hip: MFMA results in VGPRs -- its kernels read them with vector instructions in the main loop)
            flags = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if name == "conv_mfma_blk.hip" else []
            jobs.append([hipcc, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17"] + flags + ["-c", src, "-o", obj])
    if not jobs and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(o) for o in objs):
        return _SO
    if verbose:
        for j in jobs:
            print(" ".join(j))
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(subprocess.check_call, jobs))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", _SO])
    return _SO


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_SO):
        raise RgbdError(f"{_SO} not found: run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); "
                        "this package has no CPU fallback")
    L = ctypes.CDLL(_SO)
    c_i32, c_i64, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
    i32p, i64p, f32p = ctypes.POINTER(c_i32), ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_float)
    u8p, u32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint32)
    u8pp = ctypes.POINTER(u8p)
    sig = {
        "rgbd_abi_version": (ctypes.c_int, []),
        "rgbd_set_blocking_sync": (ctypes.c_int, [c_i32]),
        "rgbd_get_blocking_sync": (ctypes.c_int, []),
        "rgbd_pmf_to_quantized_cdf": (ctypes.c_int, [f32p, c_i32, c_i32, u32p]),
        "rgbd_tables_create": (ctypes.c_int, [i32p, c_i32, i32p, i32p, c_i32, ctypes.POINTER(c_vp)]),
        "rgbd_tables_destroy": (None, [c_vp]),
        "rgbd_rans_max_bytes": (c_i64, [c_i64]),
        "rgbd_rans_encode": (ctypes.c_int, [c_vp, i32p, i32p, c_i64, u8p, c_i64, i64p]),
        "rgbd_rans_decoder_create": (ctypes.c_int, [ctypes.POINTER(c_vp)]),
        "rgbd_rans_decoder_set_stream": (ctypes.c_int, [c_vp, u8p, c_i64]),
        "rgbd_rans_decoder_decode": (ctypes.c_int, [c_vp, c_vp, i32p, c_i64, i32p]),
        "rgbd_rans_decoder_destroy": (None, [c_vp]),
        "rgbd_rans_encode_batch_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp]),
        "rgbd_rans_decode_batch_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
        "rgbd_ckbd_quant_index": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, f32p, c_vp, c_vp, c_vp, c_vp]),
        "rgbd_ckbd_dequant": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
        "rgbd_conv2d_nchw": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, f32p, f32p, c_i32, c_i32, c_i32, c_i32,
                                            c_i32, c_i32, c_vp, c_vp, c_vp]),
        "rgbd_conv2d_ref_nchw": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, f32p, f32p, c_i32, c_i32, c_i32, c_i32,
                                                c_i32, c_i32, c_vp, c_vp, c_vp, i32p, c_i32, c_i32, c_i32]),
        "rgbd_pointwise_nchw": (ctypes.c_int, [c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, f32p, f32p, c_vp, c_vp]),
        "rgbd_elic_create": (ctypes.c_int, [c_i32, c_i32, i32p, c_i32, ctypes.POINTER(c_vp)]),
        "rgbd_elic_destroy": (None, [c_vp]),
        "rgbd_elic_set_ref_blocks": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, i32p, c_i32]),
        "rgbd_elic_get_refnum": (ctypes.c_int, [c_vp]),
        "rgbd_elic_ref_table_misses": (ctypes.c_int, [c_vp]),
        "rgbd_elic_clone_shared": (ctypes.c_int, [c_vp, ctypes.POINTER(c_vp)]),
        "rgbd_elic_set_tensor": (ctypes.c_int, [c_vp, ctypes.c_char_p, f32p, i64p, c_i32]),
        "rgbd_elic_set_tables": (ctypes.c_int, [c_vp, c_i32, i32p, c_i32, i32p, i32p, c_i32]),
        "rgbd_elic_set_scale_table": (ctypes.c_int, [c_vp, f32p, c_i32]),
        "rgbd_elic_finalize": (ctypes.c_int, [c_vp]),
        "rgbd_elic_compress": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
        "rgbd_elic_forward": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
        "rgbd_elic_stream_count": (ctypes.c_int, [c_vp, c_i32, c_i32]),
        "rgbd_elic_stream": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, ctypes.POINTER(u8p), i64p]),
        "rgbd_elic_decompress": (ctypes.c_int, [c_vp, u8pp, i64p, c_i32, u8pp, i64p, u8pp, i64p, u8pp, i64p, c_i32,
                                                c_i32, c_i32, c_vp, c_vp, c_vp]),
        "rgbd_elic_compress_united": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
        "rgbd_elic_decompress_united": (ctypes.c_int, [c_vp, u8pp, i64p, c_i32, u8pp, i64p, c_vp, c_vp, c_i32, c_i32,
                                                       c_i32, c_vp, c_vp, c_vp]),
        "rgbd_elic_create_r2d": (ctypes.c_int, [c_i32, c_i32, i32p, c_i32, ctypes.POINTER(c_vp)]),
        "rgbd_elic_create_stf": (ctypes.c_int, [c_i32, c_i32, i32p, c_i32, ctypes.POINTER(c_vp)]),
        "rgbd_elic_create_single": (ctypes.c_int, [c_i32, c_i32, i32p, c_i32, c_i32, ctypes.POINTER(c_vp)]),
        "rgbd_elic_compress_single": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
        "rgbd_elic_forward_single": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
        "rgbd_elic_decompress_single": (ctypes.c_int, [c_vp, u8pp, i64p, c_i32, u8pp, i64p, c_i32, c_i32, c_i32, c_vp, c_vp]),
        "rgbd_elic_debug_tensor": (ctypes.c_int, [c_vp, ctypes.c_char_p, f32p, c_i64, i32p]),
        "rgbd_elic_debug_symbols": (ctypes.c_int, [c_vp, c_i32, i32p, i32p, c_i64, i64p]),
        "rgbd_elic_set_profile": (ctypes.c_int, [c_vp, c_i32]),
        "rgbd_elic_set_debug_floats": (ctypes.c_int, [c_vp, c_i32]),
        "rgbd_elic_debug_floats": (ctypes.c_int, [c_vp, c_i32, f32p, f32p, c_i64, i64p]),
        "rgbd_elic_set_forced_symbols": (ctypes.c_int, [c_vp, c_i32, i32p, c_i64, i32p, c_i64]),
        "rgbd_elic_graph_count": (ctypes.c_int, [c_vp]),
        "rgbd_elic_workspace_bytes": (c_i64, [c_vp]),
        "rgbd_msssim_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
        "rgbd_msssim_stats": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, f32p, ctypes.c_float, c_i32, c_vp, c_vp, c_i64, c_vp]),
        "rgbd_debug_force_splitk": (ctypes.c_int, [c_i32]),
        "rgbd_debug_force_ckbd": (ctypes.c_int, [c_i32]),
        "rgbd_debug_force_blocked": (ctypes.c_int, [c_i32]),
        "rgbd_debug_force_fuse": (ctypes.c_int, [c_i32]),
        "rgbd_debug_force_subpix": (ctypes.c_int, [c_i32]),
        "rgbd_layernorm": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp]),
        "rgbd_debug_force_layernorm_form": (None, [c_i32]),
        "rgbd_debug_force_pair": (ctypes.c_int, [c_i32]),
        "rgbd_debug_fail_captures": (ctypes.c_int, [c_i32]),
        "rgbd_debug_bench_streams": (ctypes.c_int, [c_i32]),
        "rgbd_elic_set_tile_mode": (ctypes.c_int, [ctypes.c_void_p, c_i32]),
        "rgbd_debug_force_tile": (ctypes.c_int, [ctypes.c_char_p]),
        "rgbd_debug_conv_log": (ctypes.c_int, [c_i32]),
        "rgbd_debug_conv_log_read": (c_i64, [ctypes.c_char_p, c_i64]),
        "rgbd_debug_tile_override": (ctypes.c_int, [ctypes.c_char_p]),
        "rgbd_conv_bench": (ctypes.c_int, [c_i32] * 11 + [f32p]),
        "rgbd_elic_profile_dump": (ctypes.c_int, [c_vp, ctypes.c_char_p]),
        "rgbd_elic_profile_read": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_double), i64p,
                                                  ctypes.POINTER(ctypes.c_double)]),
        "rgbd_elic_profile_read_executed": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export what include/rgbd_amd.h declares
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


EXPORTS = ["rgbd_abi_version", "rgbd_set_blocking_sync", "rgbd_get_blocking_sync", "rgbd_pmf_to_quantized_cdf", "rgbd_tables_create", "rgbd_tables_destroy",
           "rgbd_rans_max_bytes", "rgbd_rans_encode", "rgbd_rans_decoder_create", "rgbd_rans_decoder_set_stream",
           "rgbd_rans_decoder_decode", "rgbd_rans_decoder_destroy", "rgbd_rans_encode_batch_dev", "rgbd_rans_decode_batch_dev", "rgbd_ckbd_quant_index", "rgbd_ckbd_dequant", "rgbd_conv2d_nchw", "rgbd_conv2d_ref_nchw", "rgbd_pointwise_nchw", "rgbd_elic_create",
           "rgbd_elic_destroy", "rgbd_elic_set_ref_blocks", "rgbd_elic_get_refnum", "rgbd_elic_ref_table_misses", "rgbd_elic_clone_shared", "rgbd_elic_set_tensor", "rgbd_elic_set_tables", "rgbd_elic_set_scale_table",
           "rgbd_elic_finalize", "rgbd_elic_compress", "rgbd_elic_forward", "rgbd_elic_stream_count", "rgbd_elic_stream",
           "rgbd_elic_decompress", "rgbd_elic_create_r2d", "rgbd_elic_create_stf", "rgbd_elic_create_single", "rgbd_elic_compress_single", "rgbd_elic_decompress_single", "rgbd_elic_forward_single", "rgbd_elic_compress_united", "rgbd_elic_decompress_united", "rgbd_elic_debug_tensor", "rgbd_elic_debug_symbols", "rgbd_elic_set_debug_floats", "rgbd_elic_debug_floats", "rgbd_elic_set_forced_symbols", "rgbd_elic_set_profile", "rgbd_elic_graph_count", "rgbd_elic_workspace_bytes", "rgbd_msssim_workspace_bytes", "rgbd_msssim_stats", "rgbd_layernorm", "rgbd_debug_force_layernorm_form",
           "rgbd_elic_profile_read", "rgbd_elic_profile_read_executed", "rgbd_debug_force_splitk", "rgbd_debug_force_fuse", "rgbd_debug_force_subpix", "rgbd_debug_force_pair", "rgbd_debug_fail_captures", "rgbd_debug_force_ckbd", "rgbd_debug_force_blocked", "rgbd_debug_bench_streams", "rgbd_elic_set_tile_mode", "rgbd_debug_force_tile", "rgbd_debug_conv_log", "rgbd_debug_conv_log_read", "rgbd_debug_tile_override", "rgbd_conv_bench",
           "rgbd_elic_profile_dump"]


_BS_HOLDERS = 0


def set_blocking_sync(on: bool) -> bool:
    """Hold (True) / release (False) the sleeping wait policy of this process's GPU (hipDeviceScheduleBlockingSync).

    Since round 5 the library itself switches a device to that policy when the FIRST engine is created on it and refuses to
    change it while any engine is alive (the round-4 `hipFree never returns` record needs work submitted under one policy and
    waited for under the other; RGBD_SPIN_WAIT=1 opts out for the whole process).  Pools and the pipelined harness therefore
    find it on; this function only counts holders -- the policy goes back to the runtime's default when the last holder lets
    go AND no engine is left -- and never raises over a refusal.  Returns whether the policy now is what was asked for."""
    global _BS_HOLDERS
    import gc

    L = lib()
    if on:
        _BS_HOLDERS += 1
        return L.rgbd_set_blocking_sync(1) == 0
    _BS_HOLDERS = max(0, _BS_HOLDERS - 1)
    if _BS_HOLDERS:
        return L.rgbd_get_blocking_sync() == 0
    gc.collect()  # engines that are already garbage go first: they are destroyed under the policy they ran with
    return L.rgbd_set_blocking_sync(0) == 0  # (refused -- and left alone -- while an engine is alive)
