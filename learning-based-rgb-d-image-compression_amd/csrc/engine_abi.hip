// Host runtime of the gfx950 codec engine, part 3 of 3: the C ABI of include/rgbd_amd.h -- engine life cycle (create /
// finalize / clone / destroy), the codec entry points, the operator-level entry points (conv2d, pointwise ops, conv
// bench) and every test / debug hook.  The coder's stand-alone entry points live in coder_abi.hip.
#include "engine.h"

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int rgbd_abi_version(void) { return RGBD_AMD_ABI_VERSION; }

// Host threads that wait for the GPU sleep instead of spinning (hipDeviceScheduleBlockingSync for the current device).
// The policy is a device flag of the process.  Work submitted under one policy and awaited under the other is what the two
// "hipFree never returns" records have in common (profiles/r03_hang_diagnosis.txt; round 4: an engine that had run under
// the spinning policy was garbage-collected -- rgbd_elic_destroy -> hipFree -> every stream of the device -- right after a
// CodecPool had switched the policy): so the device is drained under the OLD policy before the flag changes, and the Python
// side collects garbage engines first (pool.py).  (Measured and dropped: switching to the spinning policy around every
// hipFree -- with a pool's other threads launching in that window it produced exactly such mixed waits, and the suite hung.)
// Wait policy of the host threads (round 5; advisor findings on the round-4 `hipFree never returns` record).  The hang needs
// work submitted under one policy and waited for under the other, so the policy no longer moves while an engine exists: the
// FIRST engine created on a device switches that device to hipDeviceScheduleBlockingSync (sleeping waits: what every pooled
// or pipelined user wants, and ~1 ms of a 190 ms call for a lone one) before it has launched anything, and
// rgbd_set_blocking_sync() refuses (RGBD_ESTATE) to change the policy while any engine is alive.  RGBD_SPIN_WAIT=1 keeps the
// runtime's default (spinning) policy for the whole process instead.
static std::atomic<int> g_live_engines{0};
static std::mutex g_policy_mu;
static bool g_policy_done[64] = {false};

static int ensure_wait_policy()
{
    static const bool spin = getenv("RGBD_SPIN_WAIT") != nullptr;
    if (spin) return RGBD_OK;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(g_policy_mu);
    if (dev < 0 || dev >= 64 || g_policy_done[dev]) return RGBD_OK;
    if (g_live_engines.load() == 0) {
        std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);
        HIP_TRY(hipDeviceSynchronize());  // (whatever the caller's framework has in flight drains under the old policy)
        HIP_TRY(hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
    }
    g_policy_done[dev] = true;
    return RGBD_OK;
}

int rgbd_get_blocking_sync(void);
int rgbd_set_blocking_sync(int32_t on)
{
    const int cur = rgbd_get_blocking_sync();
    if (cur < 0) return cur;
    if ((on ? 1 : 0) == cur) return RGBD_OK;
    if (g_live_engines.load() > 0) return RGBD_ESTATE;  // never under a live engine's feet
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // a device-wide wait: not while a stream of this process captures
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipSetDeviceFlags(on ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto));
    if (!on) {
        std::lock_guard<std::mutex> g(g_policy_mu);
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) g_policy_done[dev] = false;  // the next first engine decides again
    }
    return RGBD_OK;
}

int rgbd_get_blocking_sync(void)
{
    unsigned flags = 0;
    HIP_TRY(hipGetDeviceFlags(&flags));
    return (flags & hipDeviceScheduleMask) == hipDeviceScheduleBlockingSync ? 1 : 0;
}

// blocked accumulation for a layer of cin_pad channels: block boundaries (in channels, multiples of 16) -> ConvArgs::blk_end
static int set_blocks(ConvArgs* a, const int32_t* blocks, int nblocks)
{
    memset(a->blk_end, 0, sizeof(a->blk_end));
    const int n16 = a->cin_pad / 16;
    if (n16 > 256) return RGBD_EINVAL;
    if (!blocks || nblocks <= 0) {  // every 16-channel chunk is a block (the multi-tap kernels of the reference's CPU library)
        for (int c = 0; c < n16; ++c) a->blk_end[c >> 5] |= 1u << (c & 31);
    } else {
        int pos = 0;
        for (int b = 0; b < nblocks; ++b) {
            if (blocks[b] <= 0 || (blocks[b] % 16 && b + 1 < nblocks)) return RGBD_EINVAL;
            pos += blocks[b];
            const int c = (pos + 15) / 16 - 1;
            if (c >= n16) return RGBD_EINVAL;
            a->blk_end[c >> 5] |= 1u << (c & 31);
        }
        if ((pos + 15) / 16 != n16) return RGBD_EINVAL;
    }
    a->blocked = 1;
    return RGBD_OK;
}

static int conv2d_nchw_impl(const float* x_dev, int32_t n, int32_t cin, int32_t h, int32_t w, const float* weight,
                            const float* bias, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t transposed,
                            int32_t act, const float* residual_dev, float* y_dev, void* stream, int refmode,
                            const int32_t* blocks, int32_t nblocks, int32_t bias_mode, int32_t flags);

int rgbd_conv2d_nchw(const float* x_dev, int32_t n, int32_t cin, int32_t h, int32_t w, const float* weight,
                     const float* bias, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t transposed,
                     int32_t act, const float* residual_dev, float* y_dev, void* stream)
{
    return conv2d_nchw_impl(x_dev, n, cin, h, w, weight, bias, cout, k, stride, pad, transposed, act, residual_dev, y_dev, stream,
                            0, nullptr, 0, 0, 0);
}

// The same layer in the reference's CPU arithmetic: channels stored permuted (rgbd_cperm), accumulation in
// blocks (`blocks`: channels per block, nullptr = one block per 16 channels), bias_mode as ConvArgs::bias_mode.
// flags bit 0: sigmoid as the reference's vector kernel computes it; bit 1: reduce the blocks as split-K ranges (1x1 layers)
int rgbd_conv2d_ref_nchw(const float* x_dev, int32_t n, int32_t cin, int32_t h, int32_t w, const float* weight,
                         const float* bias, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t transposed,
                         int32_t act, const float* residual_dev, float* y_dev, void* stream, const int32_t* blocks,
                         int32_t nblocks, int32_t bias_mode, int32_t flags)
{
    return conv2d_nchw_impl(x_dev, n, cin, h, w, weight, bias, cout, k, stride, pad, transposed, act, residual_dev, y_dev, stream,
                            1, blocks, nblocks, bias_mode, flags);
}

static int conv2d_nchw_impl(const float* x_dev, int32_t n, int32_t cin, int32_t h, int32_t w, const float* weight,
                            const float* bias, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t transposed,
                            int32_t act, const float* residual_dev, float* y_dev, void* stream, int refmode,
                            const int32_t* blocks, int32_t nblocks, int32_t bias_mode, int32_t flags)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!x_dev || !weight || !y_dev || n <= 0 || cin <= 0 || cout <= 0 || k <= 0 || k > 5 || stride < 1 || stride > 2)
        return RGBD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    HostTensor hw, hb;
    hw.shape = transposed ? std::vector<int64_t>{cin, cout, k, k} : std::vector<int64_t>{cout, cin, k, k};
    hw.v.assign(weight, weight + (size_t)cin * cout * k * k);
    if (bias) {
        hb.shape = {cout};
        hb.v.assign(bias, bias + cout);
    }
    PackedConv pc;
    const bool subpix = !refmode && transposed && g_subpix == 2 && cout <= 4 && k == 5 && stride == 2 && pad == 2 && !residual_dev;
    const int perm = refmode ? 1 : 0;
    int rc = subpix ? pack_subpix(hw, bias ? &hb : nullptr, &pc, nullptr)
                    : pack_conv(hw, bias ? &hb : nullptr, transposed != 0, &pc, nullptr, perm, perm);
    if (rc) return rc;
    int OH, OW;
    if (!transposed) {
        OH = (h + 2 * pad - k) / stride + 1;
        OW = (w + 2 * pad - k) / stride + 1;
    } else {
        OH = (h - 1) * stride - 2 * pad + k + (stride - 1);
        OW = (w - 1) * stride - 2 * pad + k + (stride - 1);
    }
    float *xin = nullptr, *yout = nullptr, *res = nullptr;
    const size_t xb = (size_t)n * h * w * pc.cin_pad * sizeof(float), yb = (size_t)n * OH * OW * pc.cout_pad * sizeof(float);
    HIP_TRY(hipMalloc((void**)&xin, xb));
    HIP_TRY(hipMalloc((void**)&yout, yb));
    rc = launch_nchw_to_nhwc16(x_dev, n, cin, h, w, xin, pc.cin_pad, s, perm);
    if (!rc && residual_dev) {
        HIP_TRY(hipMalloc((void**)&res, yb));
        rc = launch_nchw_to_nhwc16(residual_dev, n, cout, OH, OW, res, pc.cout_pad, s, perm);
    }
    if (!rc) {
        ConvArgs a{};
        a.x = xin;
        a.N = n;
        a.H = h;
        a.W = w;
        a.xcs = pc.cin_pad;
        a.cin_pad = pc.cin_pad;
        a.w = pc.w;
        a.ntaps_total = subpix ? 9 : k * k;
        a.bias = pc.bias;
        a.y = yout;
        a.OH = OH;
        a.OW = OW;
        a.ycs = pc.cout_pad;
        a.cout_pad = pc.cout_pad;
        if (subpix) {
            make_taps_subpix(&a);
            HIP_TRY(hipMemsetAsync(yout, 0, yb, s));  // channels 4..15 are not written in this form
        } else {
            make_taps(pc, stride, pad, &a);
        }
        a.GH = transposed ? h : OH;
        a.GW = transposed ? w : OW;
        a.act = act;
        if (res) {
            a.res1 = res;
            a.r1cs = pc.cout_pad;
        }
        float* part = nullptr;
        a.splitk = (g_force_splitk > 0 && !refmode) ? std::min(g_force_splitk, pc.cin_pad / 16) : 1;
        if (refmode) {
            a.bias_mode = bias_mode;
            a.exact_math = flags & 1;
            if ((flags & 2) && blocks && nblocks > 1 && nblocks <= 16) {  // the blocks as split-K ranges of the single-chain kernel
                a.splitk = nblocks;
                int pos = 0;
                for (int b = 0; b < nblocks; ++b) {
                    a.split_c16[b] = (uint16_t)(pos / 16);
                    pos += blocks[b];
                }
                a.split_c16[nblocks] = (uint16_t)((pos + 15) / 16);
                if (bias_mode == 1) rc = RGBD_EINVAL;
            } else if (blocks && nblocks == 1) {
                if (bias_mode == 1) bias_mode = a.bias_mode = 0, rc = RGBD_OK;  // (one block: S_0 + bias is the epilogue's add)
            } else {
                rc = set_blocks(&a, blocks, nblocks);
            }
        }
        if (a.splitk > 1) {
            HIP_TRY(hipMalloc((void**)&part, (size_t)a.splitk * yb));
            a.partial = part;
        }
        a.ckbd = g_force_ckbd;
        if (a.ckbd) HIP_TRY(hipMemsetAsync(yout, 0, yb, s));  // the half that is not computed reads as zero
        if (!rc) rc = launch_conv(a, s);
        if (part) {
            (void)hipStreamSynchronize(s);
            (void)hipFree(part);
        }
    }
    if (!rc) rc = launch_nhwc_to_nchw_clamp(yout, n, cout, OH, OW, pc.cout_pad, y_dev, 0, s, perm);
    hipError_t e = hipStreamSynchronize(s);
    if (!rc && e != hipSuccess) rc = RGBD_EHIP;
    (void)hipFree(xin);
    (void)hipFree(yout);
    (void)hipFree(res);
    (void)hipFree(pc.w);
    (void)hipFree(pc.bias);
    return rc;
}

// Pointwise operators of Bi-SPF / ESA / SE_Block alone (modules/transform/attention.py:52-97), NCHW in, NCHW out: what
// tests/test_gpu_pointwise.py compares with torch's F.max_pool2d / F.interpolate / the SE_Block arithmetic.
//   op 0: max_pool2d(kernel 7, stride 3) -> y [n, c, (h-7)/3+1, (w-7)/3+1]
//   op 1: interpolate(bilinear, align_corners=False) to (oh, ow)
//   op 2: SE_Block: y = x * sigmoid(fc2(relu(fc0(mean_hw(x)))))   (w0 [c/16][c], w1 [c][c/16], no biases)
//   op 3: the engine's residual form x + x * gate (entropy.py:75)
int rgbd_pointwise_nchw(int32_t op, const float* x_dev, int32_t n, int32_t c, int32_t h, int32_t w, int32_t oh, int32_t ow,
                        const float* w0, const float* w1, float* y_dev, void* stream)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!x_dev || !y_dev || n <= 0 || c <= 0 || h <= 0 || w <= 0 || op < 0 || op > 3) return RGBD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int cs = round_up(c, 16);
    if (op == 0) {
        oh = (h - 7) / 3 + 1;
        ow = (w - 7) / 3 + 1;
        if (h < 7 || w < 7) return RGBD_EINVAL;
    } else if (op >= 2) {
        oh = h;
        ow = w;
        if (!w0 || !w1 || c % 16) return RGBD_EINVAL;
    } else if (oh <= 0 || ow <= 0) {
        return RGBD_EINVAL;
    }
    // (test / tool entry point, not the codec path.)  Every buffer is released on every way out.
    struct Bufs {
        float *xin = nullptr, *yout = nullptr, *aux = nullptr, *dw0 = nullptr, *dw1 = nullptr;
        ~Bufs()
        {
            (void)hipFree(xin);
            (void)hipFree(yout);
            (void)hipFree(aux);
            (void)hipFree(dw0);
            (void)hipFree(dw1);
        }
    } b;
    float *&xin = b.xin, *&yout = b.yout, *&aux = b.aux, *&dw0 = b.dw0, *&dw1 = b.dw1;
    HIP_TRY(hipMalloc((void**)&xin, (size_t)n * h * w * cs * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&yout, (size_t)n * oh * ow * cs * sizeof(float)));
    // (the operators as the codec runs them: channels stored permuted, the reference's CPU arithmetic -- DESIGN.md 4a)
    int rc = launch_nchw_to_nhwc16(x_dev, n, c, h, w, xin, cs, s, 1);
    if (!rc && op == 0) rc = launch_maxpool7s3(xin, n, h, w, cs, yout, oh, ow, s);
    if (!rc && op == 1) rc = launch_bilinear(xin, n, h, w, cs, yout, oh, ow, s, nullptr, nullptr, c);
    if (!rc && op >= 2) {
        const int hid = c / 16;
        HIP_TRY(hipMalloc((void**)&aux, (size_t)n * (2 * c + hid + 1) * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&dw0, (size_t)c * hid * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&dw1, (size_t)c * hid * sizeof(float)));
        HIP_TRY(hipMemcpy(dw0, w0, (size_t)c * hid * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dw1, w1, (size_t)c * hid * sizeof(float), hipMemcpyHostToDevice));
        float *mean = aux, *sc = aux + (size_t)n * c, *hd = aux + (size_t)2 * n * c;
        rc = launch_channel_mean_ref(xin, n, h * w, cs, c, mean, c, s);
        if (!rc) rc = launch_se_fc_ref(mean, n, c, hid, dw0, dw1, nullptr, nullptr, hd, sc, s);  // (every row in the main order)
        if (!rc) rc = launch_channel_scale_to(xin, n, h * w, cs, c, sc, op == 3 ? 1 : 0, yout, cs, s);
    }
    if (!rc) rc = launch_nhwc_to_nchw_clamp(yout, n, c, oh, ow, cs, y_dev, 0, s, 1);
    const hipError_t e = hipStreamSynchronize(s);
    if (!rc && e != hipSuccess) rc = RGBD_EHIP;
    return rc;
}

int rgbd_debug_force_tile(const char* cfg)
{
    snprintf(g_conv_force, sizeof(g_conv_force), "%s", cfg ? cfg : "");
    ++g_cfg_epoch;
    return RGBD_OK;
}

int rgbd_debug_tile_override(const char* csv)
{
    ++g_cfg_epoch;  // cached HIP graphs have the old kernel choices baked in
    return conv_tile_override(csv);
}

int rgbd_debug_conv_log(int32_t on) { return conv_log_enable(on); }
int64_t rgbd_debug_conv_log_read(char* buf, int64_t cap) { return conv_log_read(buf, (long)cap); }

int rgbd_elic_set_tile_mode(rgbd_elic* m, int32_t mode)
{
    if (!m || mode < 0 || mode > 1) return RGBD_EINVAL;
    m->tile_mode = mode;
    return RGBD_OK;
}

int rgbd_debug_bench_streams(int32_t n)
{
    if (n < 1 || n > 32) return RGBD_EINVAL;
    g_bench_streams = n;
    return RGBD_OK;
}

int rgbd_debug_force_blocked(int32_t on)
{
    g_force_blocked = on ? 1 : 0;
    return RGBD_OK;
}

int rgbd_debug_force_ckbd(int32_t part)
{
    if (part < 0 || part > 2) return RGBD_EINVAL;
    g_force_ckbd = part;
    ++g_cfg_epoch;
    return RGBD_OK;
}

// -1: fuse where the plan says so (default), 0: never, 1 / 2 / 4: always, with 64 / 128 / 256-pixel tiles
int rgbd_debug_force_fuse(int32_t mode)
{
    const int lead_off = mode >= 15 ? 1 : 0;  // + 16: tails only, the next block's leading 1x1 stays a launch of its own
    if (lead_off) mode -= 16;
    if (mode != -1 && mode != 0 && mode != 1 && mode != 2 && mode != 4) return RGBD_EINVAL;
    g_fuse_force = mode;
    g_fuse_lead_off = lead_off;
    ++g_cfg_epoch;
    return RGBD_OK;
}

// 0: per-phase form, 1: sub-pixel form inside the codec (default), 2: also in rgbd_conv2d_nchw (tests)
int rgbd_debug_fail_captures(int32_t n)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);
    if (n < 0) return RGBD_EINVAL;
    g_fail_captures = n;
    return RGBD_OK;
}

int rgbd_debug_force_pair(int32_t mode)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);
    if (mode != 0 && mode != 1) return RGBD_EINVAL;
    g_pair = mode;
    ++g_cfg_epoch;
    return RGBD_OK;
}

int rgbd_debug_force_subpix(int32_t mode)
{
    if (mode < 0 || mode > 2) return RGBD_EINVAL;
    g_subpix = mode;
    ++g_cfg_epoch;
    return RGBD_OK;
}

int rgbd_debug_force_splitk(int32_t s)
{
    g_force_splitk = s;
    ++g_cfg_epoch;
    return RGBD_OK;
}

// Kernel-only timing of one conv shape on NHWC buffers (tools/conv_sweep.py); not part of the codec path.
int rgbd_conv_bench(int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t k, int32_t stride, int32_t pad,
                    int32_t transposed, int32_t with_residual, int32_t iters, float* ms_out)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!ms_out || n <= 0 || iters <= 0) return RGBD_EINVAL;
    HostTensor hw;
    hw.shape = transposed ? std::vector<int64_t>{cin, cout, k, k} : std::vector<int64_t>{cout, cin, k, k};
    hw.v.assign((size_t)cin * cout * k * k, 0.01f);
    PackedConv pc;
    int rc = pack_conv(hw, nullptr, transposed != 0, &pc);
    if (rc) return rc;
    int OH, OW;
    if (!transposed) {
        OH = (h + 2 * pad - k) / stride + 1;
        OW = (w + 2 * pad - k) / stride + 1;
    } else {
        OH = (h - 1) * stride - 2 * pad + k + (stride - 1);
        OW = (w - 1) * stride - 2 * pad + k + (stride - 1);
    }
    float *x = nullptr, *y = nullptr, *r = nullptr;
    const size_t xb = (size_t)n * h * w * pc.cin_pad * sizeof(float), yb = (size_t)n * OH * OW * pc.cout_pad * sizeof(float);
    HIP_TRY(hipMalloc((void**)&x, xb));
    HIP_TRY(hipMalloc((void**)&y, yb));
    HIP_TRY(hipMemset(x, 0x3c, xb));  // small positive floats
    if (with_residual) {
        HIP_TRY(hipMalloc((void**)&r, yb));
        HIP_TRY(hipMemset(r, 0x3c, yb));
    }
    ConvArgs a{};
    a.x = x;
    a.N = n;
    a.H = h;
    a.W = w;
    a.xcs = pc.cin_pad;
    a.cin_pad = pc.cin_pad;
    a.w = pc.w;
    a.ntaps_total = k * k;
    a.bias = pc.bias;
    a.y = y;
    a.OH = OH;
    a.OW = OW;
    a.ycs = pc.cout_pad;
    a.cout_pad = pc.cout_pad;
    make_taps(pc, stride, pad, &a);
    a.GH = transposed ? h : OH;
    a.GW = transposed ? w : OW;
    a.act = ACT_RELU;
    if (r) {
        a.res1 = r;
        a.r1cs = pc.cout_pad;
    }
    float* part = nullptr;
    {
        int mt = 1;
        for (int ph = 0; ph < a.nphase; ++ph) mt = std::max(mt, (int)a.taps.n[ph]);
        a.splitk = g_force_splitk > 0 ? std::min(g_force_splitk, pc.cin_pad / 16)
                                      : (g_force_splitk < 0 ? conv_splitk_for(pc.cin_pad, pc.cout_pad, mt, (long)OH * OW, a.nphase) : 1);
        if (a.splitk > 1) {
            HIP_TRY(hipMalloc((void**)&part, (size_t)a.splitk * yb));
            a.partial = part;
        }
    }
    a.ckbd = g_force_ckbd;
    a.loaded = g_bench_streams > 1 ? 1 : 0;
    if (g_force_blocked && a.splitk == 1) {  // blocked accumulation: a block per 16 channels (multi-tap) / per 96 (1x1)
        std::vector<int32_t> bl;
        if (k == 1)
            for (int c = 0; c < pc.cin_pad; c += 96) bl.push_back(std::min(96, pc.cin_pad - c));
        rc = set_blocks(&a, bl.empty() ? nullptr : bl.data(), (int)bl.size());
        a.bias_mode = k == 1 ? 2 : (transposed ? 0 : 1);
        if (rc) return rc;
    }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    rc = launch_conv(a, nullptr);
    HIP_TRY(hipDeviceSynchronize());
    float ms = 0.f;
    if (g_bench_streams <= 1) {
        HIP_TRY(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters && !rc; ++i) rc = launch_conv(a, nullptr);
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms / iters;
    } else {
        // loaded mode (rgbd_debug_bench_streams): the same launch on S streams at once -- what a launch costs in CU time
        // when the chip is shared with other engine instances, the regime the job throughput is measured in
        const int S = g_bench_streams;
        std::vector<hipStream_t> st(S);
        std::vector<hipEvent_t> done(S);
        for (int k = 0; k < S; ++k) {
            HIP_TRY(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(e0, nullptr));
        for (int k = 0; k < S; ++k) HIP_TRY(hipStreamWaitEvent(st[k], e0, 0));
        for (int i = 0; i < iters && !rc; ++i)
            for (int k = 0; k < S && !rc; ++k) rc = launch_conv(a, st[k]);
        for (int k = 0; k < S; ++k) {
            HIP_TRY(hipEventRecord(done[k], st[k]));
            HIP_TRY(hipStreamWaitEvent(nullptr, done[k], 0));
        }
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms / (iters * S);
        for (int k = 0; k < S; ++k) {
            (void)hipEventDestroy(done[k]);
            (void)hipStreamDestroy(st[k]);
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(x);
    (void)hipFree(y);
    (void)hipFree(r);
    (void)hipFree(part);
    (void)hipFree(pc.w);
    (void)hipFree(pc.bias);
    return rc;
}

// ---- codec ------------------------------------------------------------------------------------
int rgbd_elic_create(int32_t N, int32_t M, const int32_t* slice_ch, int32_t n_slices, rgbd_elic** out)
{
    if (!out || !slice_ch || n_slices <= 0 || N % 16 || M % 16) return RGBD_EINVAL;
    int sum = 0;
    for (int i = 0; i < n_slices; ++i) {
        if (slice_ch[i] <= 0 || slice_ch[i] % 8) return RGBD_EINVAL;  // 16-byte channel views; STF_united has 24-wide slices
        sum += slice_ch[i];
    }
    if (sum != M) return RGBD_EINVAL;
    if (const int pr = ensure_wait_policy()) return pr;
    rgbd_elic* m = new rgbd_elic();
    m->N = N;
    m->M = M;
    m->slice_ch.assign(slice_ch, slice_ch + n_slices);
    *out = m;
    ++g_live_engines;
    return RGBD_OK;
}

static int check_ready(const rgbd_elic* m);

int rgbd_elic_create_r2d(int32_t N, int32_t M, const int32_t* slice_ch, int32_t n_slices, rgbd_elic** out)
{
    const int r = rgbd_elic_create(N, M, slice_ch, n_slices, out);
    if (r) return r;
    (*out)->variant = 3;
    return RGBD_OK;
}

int rgbd_elic_create_stf(int32_t N, int32_t M, const int32_t* slice_ch, int32_t n_slices, rgbd_elic** out)
{
    if (M != 384) return RGBD_EINVAL;  // embed_dim 48 * 8 (models/stf_united.py:640)
    const int r = rgbd_elic_create(N, M, slice_ch, n_slices, out);
    if (r) return r;
    (*out)->variant = 2;
    (*out)->refnum = false;  // (Swin transforms: channel slices that are not 16-aligned; the single-chain arithmetic of rounds 1-4)
    return RGBD_OK;
}

int rgbd_elic_create_single(int32_t N, int32_t M, const int32_t* slice_ch, int32_t n_slices, int32_t in_ch, rgbd_elic** out)
{
    if (in_ch < 1 || in_ch > 16) return RGBD_EINVAL;
    const int r = rgbd_elic_create(N, M, slice_ch, n_slices, out);
    if (r) return r;
    (*out)->variant = 1;
    (*out)->in_ch = in_ch;
    return RGBD_OK;
}

int rgbd_elic_compress_single(rgbd_elic* m, const float* x_dev, int32_t B, int32_t H, int32_t W, int32_t per_image_streams,
                              void* stream)
{
    int r = check_ready(m);
    if (r) return r;
    if (m->variant != 1 || !x_dev || B <= 0 || H <= 0 || W <= 0 || H % 64 || W % 64) return RGBD_EINVAL;
    if (const int ur = m->use_stream(stream)) return ur;
    const int per_image = (per_image_streams || B == 1) ? 1 : 0;
    char key[96];
    snprintf(key, sizeof(key), "c1|%d|%d|%d|%d", B, H, W, per_image);
    r = run_sized(m, key, [&]() { return m->run_compress1(x_dev, B, H, W, per_image); });
    if (m->profile) m->profile_collect();
    return r;
}

int rgbd_elic_forward_single(rgbd_elic* m, const float* x_dev, int32_t B, int32_t H, int32_t W, float* xhat_dev, float* lik_y,
                             float* lik_z, void* stream)
{
    int r = check_ready(m);
    if (r) return r;
    if (m->variant != 1 || !x_dev || !xhat_dev || !lik_y || !lik_z || B <= 0 || H <= 0 || W <= 0 || H % 64 || W % 64)
        return RGBD_EINVAL;
    if (const int ur = m->use_stream(stream)) return ur;
    char key[96];
    snprintf(key, sizeof(key), "f1|%d|%d|%d", B, H, W);
    return run_sized(m, key, [&]() { return m->run_forward1(x_dev, B, H, W, xhat_dev, lik_y, lik_z); });
}

int rgbd_elic_decompress_single(rgbd_elic* m, const uint8_t* const* y, const int64_t* y_len, int32_t n_y,
                                const uint8_t* const* z, const int64_t* z_len, int32_t B, int32_t zh, int32_t zw,
                                float* x_dev, void* stream)
{
    int r = check_ready(m);
    if (r) return r;
    if (m->variant != 1 || !y || !y_len || !z || !z_len || !x_dev || B <= 0 || zh <= 0 || zw <= 0) return RGBD_EINVAL;
    if (n_y != 1 && n_y != B) return RGBD_EINVAL;
    if (const int ur = m->use_stream(stream)) return ur;
    char key[96];
    snprintf(key, sizeof(key), "d1|%d|%d|%d|%d", B, zh, zw, n_y);
    r = run_sized(m, key, [&]() { return m->run_decompress1(y, y_len, n_y, z, z_len, B, zh, zw, x_dev); });
    if (!r) r = m->wait_stream();  // (the work may sit on the engine's own stream: return when x_hat is there)
    if (m->profile && !r) m->profile_collect();
    return r;
}

int rgbd_elic_clone_shared(const rgbd_elic* src, rgbd_elic** out)
{
    if (!src || !out || !src->finalized) return RGBD_EINVAL;
    rgbd_elic* m = new rgbd_elic();
    m->N = src->N;
    m->M = src->M;
    m->slice_ch = src->slice_ch;
    m->variant = src->variant;
    m->in_ch = src->in_ch;
    m->refnum = src->refnum;
    m->ref_threads = src->ref_threads;
    m->ref_tab = src->ref_tab;
    m->convs = src->convs;    // device pointers are shared, read-only; the generations below keep them alive
    m->dense = src->dense;
    m->gen_w = src->gen_w;
    for (int i = 0; i < 4; ++i) m->tables[i] = src->tables[i];
    m->scale_table = src->scale_table;
    m->gen_scale = src->gen_scale;
    m->finalized = true;
    m->is_clone = true;
    *out = m;
    ++g_live_engines;
    return RGBD_OK;
}

void rgbd_elic_destroy(rgbd_elic* m)
{
    const bool dbg = g_dbg_destroy;
    if (dbg) fprintf(stderr, "[destroy %p] wait lock\n", (void*)m);
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!m) return;
    if (dbg) fprintf(stderr, "[destroy %p] locked, graphs %zu\n", (void*)m, m->graphs.size());
    // weights, tables and the scale table belong to shared generations (DevGen) that go when their last user does
    m->graphs_invalidate();
    if (dbg) fprintf(stderr, "[destroy %p] graphs gone\n", (void*)m);
    if (dbg) {
        std::lock_guard<std::mutex> g(g_live_mu);
        g_live_streams.erase(m);
    }
    {
        HangWatch w("hipFree(arena) in rgbd_elic_destroy", 20, true);
        if (m->arena.base) (void)hipFree(m->arena.base);
    }
    if (dbg) fprintf(stderr, "[destroy %p] arena freed\n", (void*)m);
    if (m->pin) (void)hipHostFree(m->pin);
    if (m->res_pin) (void)hipHostFree(m->res_pin);
    if (m->pin_ev) (void)hipEventDestroy(m->pin_ev);
    if (m->done_ev) (void)hipEventDestroy(m->done_ev);
    for (hipEvent_t e : m->ev_pool) (void)hipEventDestroy(e);
    if (dbg) fprintf(stderr, "[destroy %p] events/streams gone\n", (void*)m);
    delete m;
    --g_live_engines;
    if (dbg) fprintf(stderr, "[destroy] done\n");
}

int rgbd_elic_set_tensor(rgbd_elic* m, const char* name, const float* data, const int64_t* shape, int32_t ndim)
{
    if (!m || !name || !data || !shape || ndim < 1 || ndim > 4) return RGBD_EINVAL;
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] <= 0) return RGBD_EINVAL;
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.v.assign(data, data + n);
    m->raw[name] = std::move(t);
    m->finalized = false;
    return RGBD_OK;
}

int rgbd_elic_set_tables(rgbd_elic* m, int32_t which, const int32_t* cdf, int32_t cdf_stride, const int32_t* cdf_sizes,
                         const int32_t* offsets, int32_t n_cdf)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!m || which < 0 || which > 3) return RGBD_EINVAL;
    TableSet fresh;
    const int r = build_tables(cdf, cdf_stride, cdf_sizes, offsets, n_cdf, &fresh);
    if (r) {
        if (fresh.blob) (void)hipFree(fresh.blob);
        return r;
    }
    fresh.hold = std::make_shared<DevGen>();
    fresh.hold->p.push_back(fresh.blob);
    m->tables[which] = fresh;  // the previous blob goes when no clone points at it any more
    m->graphs_invalidate();
    return RGBD_OK;
}

int rgbd_elic_set_scale_table(rgbd_elic* m, const float* table, int32_t n)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!m || !table || n != 64) return RGBD_EINVAL;
    float* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 64 * sizeof(float)));
    auto g = std::make_shared<DevGen>();
    g->p.push_back(d);
    HIP_TRY(hipMemcpy(d, table, 64 * sizeof(float), hipMemcpyHostToDevice));
    m->scale_table = d;
    m->gen_scale = g;
    m->graphs_invalidate();
    return RGBD_OK;
}

static bool ends_with(const std::string& s, const char* suf)
{
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int rgbd_elic_set_ref_blocks(rgbd_elic* m, int32_t kind, int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t batch,
                             const int32_t* blocks, int32_t nblocks)
{
    if (m && kind == 4 && blocks && nblocks == 1 && blocks[0] >= 1 && blocks[0] <= 4096) {  // CPU threads of the reference run
        m->ref_threads = blocks[0];
        m->graphs_invalidate();
        return RGBD_OK;
    }
    if (!m || !blocks || nblocks <= 0 || nblocks > (kind >= 2 ? 65536 : 64) || kind < 0 || kind > 3) return RGBD_EINVAL;
    int sum = 0;
    for (int i = 0; i < nblocks; ++i) {
        if (kind == 3 ? blocks[i] < 0 : (kind == 2 ? (blocks[i] < 0 || blocks[i] > 2) : blocks[i] <= 0)) return RGBD_EINVAL;
        sum += blocks[i];
    }
    if (kind == 0 && sum != cin) return RGBD_EINVAL;
    if (kind == 1 && nblocks > 16) return RGBD_EINVAL;
    if (kind == 2 && nblocks != cout) return RGBD_EINVAL;  // (one class per output row)
    m->ref_tab->blocks[{kind, cin, cout, h, w, batch}] = std::vector<int>(blocks, blocks + nblocks);
    m->graphs_invalidate();
    return RGBD_OK;
}

int rgbd_elic_get_refnum(const rgbd_elic* m) { return m ? (m->refnum ? 1 : 0) : RGBD_EINVAL; }
int rgbd_elic_ref_table_misses(const rgbd_elic* m) { return m ? m->ref_tab->misses : RGBD_EINVAL; }

int rgbd_elic_finalize(rgbd_elic* m)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!m) return RGBD_EINVAL;
    // a shared-weight clone has no host tensors of its own: re-finalising it would only drop the weights it borrows
    if (m->is_clone && m->raw.empty()) return RGBD_ESTATE;
    // the new weights are packed into a fresh generation and swapped in at the end; the old generation is released
    // here but lives on for as long as a clone still points into it (no use-after-free between parent and clones)
    auto gen = std::make_shared<DevGen>();
    std::map<std::string, PackedConv> convs;
    std::map<std::string, float*> dense;
    auto dev_copy = [&](const float* src, size_t n, float** out) -> int {
        float* d = nullptr;
        HIP_TRY(hipMalloc((void**)&d, n * sizeof(float)));
        gen->p.push_back(d);
        HIP_TRY(hipMemcpy(d, src, n * sizeof(float), hipMemcpyHostToDevice));
        *out = d;
        return RGBD_OK;
    };
    for (auto& kv : m->raw) {
        const std::string& name = kv.first;
        const HostTensor& t = kv.second;
        if (ends_with(name, ".weight") && t.shape.size() == 4) {
            // ConvTranspose2d layers of this model: g_s stages 1/6/12/17 and the h_s deconvs
            // (single-modal ELIC: g_s stages 1/5/10/14 and h_s.increase.*; stage numbers that are not a bare
            //  "<stage>.weight" in the other variant belong to blocks with sub-names, so the union is unambiguous)
            bool transposed = name.find(".deconv.") != std::string::npos || name.rfind("h_s.increase.", 0) == 0;
            if (name.rfind("g_s.", 0) == 0)
                for (const char* suf : {"_transform.1.weight", "_transform.6.weight", "_transform.12.weight",
                                        "_transform.17.weight", "_transform.5.weight", "_transform.10.weight",
                                        "_transform.14.weight"})
                    transposed = transposed || ends_with(name, suf);
            const std::string bname = name.substr(0, name.size() - 6) + "bias";
            auto bit = m->raw.find(bname);
            PackedConv pc;
            // refnum: activations store their channels permuted (rgbd_cperm) -- except the network's input (read channel by
            // channel by the K-packing gather) and its output images (<= 4 channels, converted straight to NCHW)
            const int pm = m->perm();
            const int k0 = (int)t.shape[2];
            const int cin0 = transposed ? (int)t.shape[0] : (int)t.shape[1], cout0 = transposed ? (int)t.shape[1] : (int)t.shape[0];
            const bool image_in = !transposed && cin0 <= 3 && k0 == 5, image_out = transposed && cout0 <= 4 && k0 == 5;
            const int r = pack_conv(t, bit == m->raw.end() ? nullptr : &bit->second, transposed, &pc, gen.get(), image_in ? 0 : pm,
                                    image_out ? 0 : pm);
            if (r) return r;
            convs[name] = pc;
            if (!transposed && pc.cin <= 3 && pc.k == 5) {  // the image-consuming layer: also as a 1x1 over a K-packed input
                PackedConv pk;
                const int r4 = pack_kpack(t, bit == m->raw.end() ? nullptr : &bit->second, &pk, gen.get(), pm);
                if (r4) return r4;
                convs[name.substr(0, name.size() - 6) + "kpack.weight"] = pk;
            }
            if (transposed && pc.cout <= 4 && pc.k == 5) {  // the image-producing layer: also in its sub-pixel form
                PackedConv ps;
                const int r3 = pack_subpix(t, bit == m->raw.end() ? nullptr : &bit->second, &ps, gen.get(), pm);
                if (r3) return r3;
                convs[name.substr(0, name.size() - 6) + "subpix.weight"] = ps;
            }
        } else if (ends_with(name, ".weight") && t.shape.size() == 2) {
            // SE_Block linears; fc.2 ([C][hidden]) is kept transposed so the gate kernel reads it coalesced
            std::vector<float> hv = t.v;
            if (m->refnum && (ends_with(name, ".fc.0.weight") || ends_with(name, ".fc.2.weight"))) {
                // the reference's Linear on one vector: per-row accumulation class, measured per layer shape
                const int J = (int)t.shape[0], K = (int)t.shape[1];
                auto ct = m->ref_tab->blocks.find({2, K, J, 0, 0, 1});
                if (ct != m->ref_tab->blocks.end() && (int)ct->second.size() == J) {
                    float* dcls = nullptr;
                    if (const int r = dev_copy(reinterpret_cast<const float*>(ct->second.data()), (size_t)J, &dcls)) return r;
                    dense[name + ".rowclass"] = dcls;
                }
            }
            if (ends_with(name, ".fc.2.weight") && !m->refnum) {
                const size_t C = (size_t)t.shape[0], Hd = (size_t)t.shape[1];
                for (size_t c = 0; c < C; ++c)
                    for (size_t j = 0; j < Hd; ++j) hv[j * C + c] = t.v[c * Hd + j];
            }
            float* d = nullptr;
            if (const int r = dev_copy(hv.data(), hv.size(), &d)) return r;
            dense[name] = d;
        } else if ((t.shape.size() == 1 && (name.find(".norm") != std::string::npos)) ||
                   ends_with(name, "relative_position_bias_table")) {
            // Swin LayerNorm affine parameters and relative position bias tables: plain device arrays
            float* d = nullptr;
            if (const int r = dev_copy(t.v.data(), t.v.size(), &d)) return r;
            dense[name] = d;
        } else if (ends_with(name, "entropy_bottleneck.quantiles")) {
            // medians = quantiles[:, 0, 1]  (entropy_models.py:316-318)
            const int C = (int)t.shape[0];
            std::vector<float> med(C);
            for (int c = 0; c < C; ++c) med[c] = t.v[(size_t)c * 3 + 1];
            float* d = nullptr;
            if (const int r = dev_copy(med.data(), (size_t)C, &d)) return r;
            dense[name.substr(0, name.size() - 9) + "medians"] = d;
            // softplus(matrix_i) / bias_i / tanh(factor_i) per channel for the eval-mode likelihood (58 floats/channel)
            const std::string pre = name.substr(0, name.size() - 9);
            std::vector<float> prm((size_t)C * 58, 0.f);
            const int fi[6] = {1, 3, 3, 3, 3, 1};
            bool ok = true;
            size_t off = 0;
            for (int i = 0; i < 5 && ok; ++i) {
                auto mi = m->raw.find(pre + "_matrix" + std::to_string(i));
                auto bi = m->raw.find(pre + "_bias" + std::to_string(i));
                auto fa = i < 4 ? m->raw.find(pre + "_factor" + std::to_string(i)) : m->raw.end();
                if (mi == m->raw.end() || bi == m->raw.end() || (i < 4 && fa == m->raw.end())) {
                    ok = false;
                    break;
                }
                const int no = fi[i + 1], ni = fi[i];
                for (int c = 0; c < C; ++c) {
                    float* p = prm.data() + (size_t)c * 58 + off;
                    for (int k = 0; k < no * ni; ++k) {
                        const float v = mi->second.v[(size_t)c * no * ni + k];
                        p[k] = v > 20.f ? v : std::log1p(std::exp(v));  // F.softplus (threshold 20)
                    }
                    for (int k = 0; k < no; ++k) p[no * ni + k] = bi->second.v[(size_t)c * no + k];
                    if (i < 4)
                        for (int k = 0; k < no; ++k) p[no * ni + no + k] = std::tanh(fa->second.v[(size_t)c * no + k]);
                }
                off += (size_t)no * ni + no + (i < 4 ? no : 0);
            }
            if (ok) {
                float* dp = nullptr;
                if (const int r = dev_copy(prm.data(), prm.size(), &dp)) return r;
                dense[pre + "cumulative"] = dp;
            }
        }
    }
    m->convs.swap(convs);
    m->dense.swap(dense);
    m->gen_w = gen;
    m->graphs_invalidate();
    m->finalized = true;
    return RGBD_OK;
}

static int check_ready(const rgbd_elic* m)
{
    if (!m || !m->finalized || !m->scale_table) return RGBD_ESTATE;
    for (int i = 0; i < 4; ++i)
        if (!m->tables[i].ready && !(m->variant == 1 && (i & 1))) return RGBD_ESTATE;  // single-modal: slots 0 and 2
    return RGBD_OK;
}

int rgbd_elic_compress(rgbd_elic* m, const float* rgb_dev, const float* depth_dev, int32_t B, int32_t H, int32_t W,
                       int32_t per_image_streams, void* stream)
{
    int r = check_ready(m);
    if (r) return r;
    if (!rgb_dev || !depth_dev || B <= 0 || H <= 0 || W <= 0 || H % 64 || W % 64) return RGBD_EINVAL;
    if (const int ur = m->use_stream(stream)) return ur;
    const int per_image = (per_image_streams || B == 1) ? 1 : 0;
    char key[96];
    snprintf(key, sizeof(key), "c|%d|%d|%d|%d", B, H, W, per_image);
    r = run_sized(m, key, [&]() { return m->run_compress(rgb_dev, depth_dev, B, H, W, per_image); });
    if (m->profile) m->profile_collect();  // run_compress ends with a stream synchronise
    return r;
}

int rgbd_elic_forward(rgbd_elic* m, const float* rgb_dev, const float* depth_dev, int32_t B, int32_t H, int32_t W,
                      float* xr_dev, float* xd_dev, float* lik_y_rgb, float* lik_y_depth, float* lik_z_rgb,
                      float* lik_z_depth, void* stream)
{
    int r = check_ready(m);
    if (r) return r;
    if (!rgb_dev || !depth_dev || !xr_dev || !xd_dev || !lik_y_rgb || !lik_y_depth || !lik_z_rgb || !lik_z_depth || B <= 0 ||
        H <= 0 || W <= 0 || H % 64 || W % 64)
        return RGBD_EINVAL;
    if (const int ur = m->use_stream(stream)) return ur;
    char key[96];
    snprintf(key, sizeof(key), "f|%d|%d|%d", B, H, W);
    return run_sized(m, key, [&]() {
        return m->run_forward(rgb_dev, depth_dev, B, H, W, xr_dev, xd_dev, lik_y_rgb, lik_y_depth, lik_z_rgb, lik_z_depth);
    });
}

int rgbd_elic_stream_count(const rgbd_elic* m, int32_t modality, int32_t kind)
{
    if (!m || modality < 0 || modality > 1 || kind < 0 || kind > 1) return RGBD_EINVAL;
    return (int)m->streams[modality][kind].size();
}

int rgbd_elic_stream(const rgbd_elic* m, int32_t modality, int32_t kind, int32_t index, const uint8_t** data,
                     int64_t* nbytes)
{
    if (!m || !data || !nbytes || modality < 0 || modality > 1 || kind < 0 || kind > 1) return RGBD_EINVAL;
    const auto& v = m->streams[modality][kind];
    if (index < 0 || index >= (int)v.size()) return RGBD_EINVAL;
    *data = v[index].data();
    *nbytes = (int64_t)v[index].size();
    return RGBD_OK;
}

int rgbd_elic_decompress(rgbd_elic* m, const uint8_t* const* y_rgb, const int64_t* y_rgb_len, int32_t n_y,
                         const uint8_t* const* y_depth, const int64_t* y_depth_len, const uint8_t* const* z_rgb,
                         const int64_t* z_rgb_len, const uint8_t* const* z_depth, const int64_t* z_depth_len, int32_t B,
                         int32_t zh, int32_t zw, float* xr_dev, float* xd_dev, void* stream)
{
    int r = check_ready(m);
    if (r) return r;
    if (!y_rgb || !y_depth || !z_rgb || !z_depth || !xr_dev || !xd_dev || B <= 0 || zh <= 0 || zw <= 0) return RGBD_EINVAL;
    if (n_y != 1 && n_y != B) return RGBD_EINVAL;
    if (const int ur = m->use_stream(stream)) return ur;
    const uint8_t* const* ys[2] = {y_rgb, y_depth};
    const int64_t* yl[2] = {y_rgb_len, y_depth_len};
    const uint8_t* const* zs[2] = {z_rgb, z_depth};
    const int64_t* zl[2] = {z_rgb_len, z_depth_len};
    char key[96];
    snprintf(key, sizeof(key), "d|%d|%d|%d|%d", B, zh, zw, n_y);
    r = run_sized(m, key, [&]() { return m->run_decompress(ys, yl, n_y, zs, zl, B, zh, zw, xr_dev, xd_dev); });
    // the caller's wait for x_hat happens here, on an event the host thread sleeps on (a pooled rank has 16 of them)
    if (!r) r = m->wait_stream();
    if (m->profile && !r) m->profile_collect();
    return r;
}

int rgbd_elic_compress_united(rgbd_elic* m, const float* y_rgb_dev, const float* hyper_rgb_dev, const float* y_depth_dev,
                              const float* hyper_depth_dev, int32_t B, int32_t h, int32_t w, int32_t per_image_streams,
                              void* stream)
{
    int r = check_ready(m);
    if (r) return r;
    if (!y_rgb_dev || !hyper_rgb_dev || !y_depth_dev || !hyper_depth_dev || B <= 0 || h <= 0 || w <= 0 || (w & 1))
        return RGBD_EINVAL;
    if (const int ur = m->use_stream(stream)) return ur;
    const int per_image = (per_image_streams || B == 1) ? 1 : 0;
    rgbd_elic::Latents lat = {{y_rgb_dev, y_depth_dev}, {hyper_rgb_dev, hyper_depth_dev}, {nullptr, nullptr}};
    char key[96];
    snprintf(key, sizeof(key), "cu|%d|%d|%d|%d", B, h, w, per_image);
    r = run_sized(m, key, [&]() { return m->run_compress(nullptr, nullptr, B, h * 16, w * 16, per_image, &lat); });
    if (m->profile) m->profile_collect();
    return r;
}

int rgbd_elic_decompress_united(rgbd_elic* m, const uint8_t* const* y_rgb, const int64_t* y_rgb_len, int32_t n_y,
                                const uint8_t* const* y_depth, const int64_t* y_depth_len, const float* hyper_rgb_dev,
                                const float* hyper_depth_dev, int32_t B, int32_t h, int32_t w, float* yhat_rgb_dev,
                                float* yhat_depth_dev, void* stream)
{
    int r = check_ready(m);
    if (r) return r;
    if (!y_rgb || !y_depth || !hyper_rgb_dev || !hyper_depth_dev || !yhat_rgb_dev || !yhat_depth_dev || B <= 0 || h <= 0 ||
        w <= 0 || (w & 1))
        return RGBD_EINVAL;
    if (n_y != 1 && n_y != B) return RGBD_EINVAL;
    if (const int ur = m->use_stream(stream)) return ur;
    const uint8_t* const* ys[2] = {y_rgb, y_depth};
    const int64_t* yl[2] = {y_rgb_len, y_depth_len};
    const uint8_t* const* zs[2] = {nullptr, nullptr};
    const int64_t* zl[2] = {nullptr, nullptr};
    rgbd_elic::Latents lat = {{nullptr, nullptr}, {hyper_rgb_dev, hyper_depth_dev}, {yhat_rgb_dev, yhat_depth_dev}};
    char key[96];
    snprintf(key, sizeof(key), "du|%d|%d|%d|%d", B, h, w, n_y);
    r = run_sized(m, key, [&]() { return m->run_decompress_impl(ys, yl, n_y, zs, zl, B, h, w, nullptr, nullptr, &lat); });
    if (!r) r = m->wait_stream();
    if (m->profile && !r) m->profile_collect();
    return r;
}

int64_t rgbd_elic_workspace_bytes(const rgbd_elic* m) { return m ? (int64_t)m->arena.cap : -1; }

int rgbd_elic_graph_count(const rgbd_elic* m)
{
    if (!m) return RGBD_EINVAL;
    int n = 0;
    for (const auto& kv : m->graphs) n += kv.second.exec ? 1 : 0;
    return n;
}

int rgbd_elic_set_profile(rgbd_elic* m, int32_t on)
{
    if (!m) return RGBD_EINVAL;
    m->profile = on != 0;
    m->profile_keys = on == 2;
    m->ev_used = 0;
    m->prof_flops = 0.0;
    m->prof_flops_exec = 0.0;
    m->prof_ms = 0.0;
    m->prof_launches = 0;
    m->prof_layers.clear();
    m->prof_counts.clear();
    m->ev_names.clear();
    return RGBD_OK;
}

int rgbd_elic_profile_dump(rgbd_elic* m, const char* path)
{
    if (!m || !path) return RGBD_EINVAL;
    FILE* f = fopen(path, "w");
    if (!f) return RGBD_EINVAL;
    fprintf(f, "layer,launches,ms,gflop,tflops,gflop_executed,tflops_executed\n");
    for (const auto& kv : m->prof_layers)
        fprintf(f, "%s,%d,%.4f,%.3f,%.2f,%.3f,%.2f\n", kv.first.c_str(), m->prof_counts[kv.first], kv.second.first,
                kv.second.second / 1e9, kv.second.first > 0 ? kv.second.second / kv.second.first / 1e9 : 0.0,
                kv.second.exec / 1e9, kv.second.first > 0 ? kv.second.exec / kv.second.first / 1e9 : 0.0);
    fclose(f);
    return RGBD_OK;
}

int rgbd_elic_profile_read(rgbd_elic* m, double* conv_ms, int64_t* launches, double* flops)
{
    if (!m || !conv_ms || !launches || !flops) return RGBD_EINVAL;
    *conv_ms = m->prof_ms;
    *launches = m->prof_launches;
    *flops = m->prof_flops;
    return RGBD_OK;
}

int rgbd_elic_profile_read_executed(rgbd_elic* m, double* flops_executed)
{
    if (!m || !flops_executed) return RGBD_EINVAL;
    *flops_executed = m->prof_flops_exec;
    return RGBD_OK;
}

int rgbd_elic_debug_tensor(rgbd_elic* m, const char* name, float* data, int64_t cap_floats, int32_t* shape_out)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!m || !name || !shape_out) return RGBD_EINVAL;
    auto it = m->named.find(name);
    if (it == m->named.end()) return RGBD_EINVAL;
    const Act& a = it->second;
    shape_out[0] = a.n;
    shape_out[1] = a.c;
    shape_out[2] = a.h;
    shape_out[3] = a.w;
    if (!data) return RGBD_OK;
    const int64_t need = (int64_t)a.n * a.c * a.h * a.w;
    if (cap_floats < need) return RGBD_ENOSPC;
    float* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, (size_t)need * sizeof(float)));
    // (x_hat tensors come out of the image-producing layers in channel order; everything else is stored permuted)
    const int pm = (m->perm() && a.c > 4) ? 1 : 0;
    int r = launch_nhwc_to_nchw_clamp(a.p, a.n, a.c, a.h, a.w, a.cs, tmp, 0, m->s, pm);
    if (!r && hipStreamSynchronize(m->s) != hipSuccess) r = RGBD_EHIP;  // (the stream may be non-blocking: hipMemcpy would not wait for it)
    if (!r && hipMemcpy(data, tmp, (size_t)need * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) r = RGBD_EHIP;
    (void)hipFree(tmp);
    return r;
}

int rgbd_elic_set_debug_floats(rgbd_elic* m, int32_t on)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);
    if (!m) return RGBD_EINVAL;
    if (m->debug_floats != (on != 0)) m->graphs_invalidate();  // the workspace layout changes
    m->debug_floats = on != 0;
    return RGBD_OK;
}

int rgbd_elic_set_forced_symbols(rgbd_elic* m, int32_t modality, const int32_t* y_sym, int64_t n_y, const int32_t* z_sym, int64_t n_z)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);
    if (!m || modality < 0 || modality > 1 || n_y < 0 || n_z < 0 || (n_y && !y_sym) || (n_z && !z_sym)) return RGBD_EINVAL;
    if (m->variant == 1) return RGBD_EINVAL;  // (the two-modality codecs only)
    m->graphs_invalidate();  // the workspace layout and the launch list change
    m->force_y[modality].assign(y_sym, y_sym + n_y);
    m->force_z[modality].assign(z_sym, z_sym + n_z);
    return RGBD_OK;
}

int rgbd_elic_debug_floats(rgbd_elic* m, int32_t modality, float* x, float* scale, int64_t cap, int64_t* n)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!m || !n || modality < 0 || modality > 1 || !m->dbg_x || !m->dbg_s) return RGBD_EINVAL;
    if (m->variant == 1 && modality != 0) return RGBD_EINVAL;  // the single-modal model keeps one modality's floats
    *n = m->dbg_per_mod;
    if (!x || !scale) return RGBD_OK;
    if (cap < m->dbg_per_mod) return RGBD_ENOSPC;
    HIP_TRY(hipStreamSynchronize(m->s));
    const size_t bytes = sizeof(float) * (size_t)m->dbg_per_mod;
    HIP_TRY(hipMemcpy(x, m->dbg_x + (size_t)modality * m->dbg_per_mod, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(scale, m->dbg_s + (size_t)modality * m->dbg_per_mod, bytes, hipMemcpyDeviceToHost));
    return RGBD_OK;
}

int rgbd_elic_debug_symbols(rgbd_elic* m, int32_t modality, int32_t* symbols, int32_t* indexes, int64_t cap, int64_t* n)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!m || !n || modality < 0 || modality > 1 || !m->dbg_sym) return RGBD_EINVAL;
    *n = m->dbg_per_mod;
    if (!symbols || !indexes) return RGBD_OK;
    if (cap < m->dbg_per_mod) return RGBD_ENOSPC;
    HIP_TRY(hipStreamSynchronize(m->s));
    HIP_TRY(hipMemcpy(symbols, m->dbg_sym + (size_t)modality * m->dbg_per_mod, sizeof(int32_t) * (size_t)m->dbg_per_mod,
                      hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(indexes, m->dbg_idx + (size_t)modality * m->dbg_per_mod, sizeof(int32_t) * (size_t)m->dbg_per_mod,
                      hipMemcpyDeviceToHost));
    return RGBD_OK;
}

}  // extern "C"
