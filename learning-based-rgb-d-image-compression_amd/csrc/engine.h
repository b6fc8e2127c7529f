// Host runtime of the gfx950 codec engine, part 1 of 3: weight packing, the workspace arena and `struct rgbd_elic` -- the
// layer graph of the four model variants (ELIC_united, single-modal ELIC, STF_united, ELIC_united_R2D) as inline methods
// that plan and issue HIP kernel launches on one stream, the conv planner (tiles, split-K, reference arithmetic) and the
// per-call-shape HIP-graph cache.  engine.hip holds the call paths (compress / decompress / forward), engine_abi.hip the
// C ABI (include/rgbd_amd.h).  Everything shared between those two translation units is `inline` here (one instance).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <dirent.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/syscall.h>
#include <thread>
#include <unistd.h>

#include "../../include/rgbd_amd.h"
#include "common.h"
#include "engine_internal.h"


// (process-wide switches and the helpers below: `inline` -- one instance for engine.hip and engine_abi.hip -- inside a namespace of
// their own, so that the weak symbols they become cannot collide with a host application's globals)
namespace rgbd_rt {


// ------------------------------------------------------------------------------------------------
// packed layers
// ------------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<float> v;
    std::vector<int64_t> shape;
};

struct PackedConv {
    float* w = nullptr;  // [cout_pad][k*k][cin_pad]
    float* bias = nullptr;
    int cin = 0, cout = 0, cin_pad = 0, cout_pad = 0, k = 0;
    bool transposed = false;
    bool subpix = false;  // pack_subpix(): [16 = phase * 4 + cout][9 taps][cin_pad] of a k = 5, stride-2 transposed conv
};

// perm_in / perm_out: store the input / output channels at their permuted positions (rgbd_cperm)
inline int pack_conv(const HostTensor& w, const HostTensor* b, bool transposed, PackedConv* pc, DevGen* gen = nullptr, int perm_in = 0,
              int perm_out = 0)
{
    if (w.shape.size() != 4 || w.shape[2] != w.shape[3]) return RGBD_EINVAL;
    const int k = (int)w.shape[2];
    const int cout = transposed ? (int)w.shape[1] : (int)w.shape[0];
    const int cin = transposed ? (int)w.shape[0] : (int)w.shape[1];
    pc->cin = cin;
    pc->cout = cout;
    pc->k = k;
    pc->transposed = transposed;
    pc->cin_pad = round_up(cin, 16);
    pc->cout_pad = round_up(cout, 16);
    const size_t n = (size_t)pc->cout_pad * k * k * pc->cin_pad;
    std::vector<float> h(n, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < k * k; ++t) {
                const size_t src = transposed ? (((size_t)ci * cout + co) * k * k + t) : (((size_t)co * cin + ci) * k * k + t);
                h[((size_t)rgbd_cperm(co, perm_out) * k * k + t) * pc->cin_pad + rgbd_cperm(ci, perm_in)] = w.v[src];
            }
    std::vector<float> hb(pc->cout_pad, 0.f);
    if (b) {
        if ((int)b->v.size() != cout) return RGBD_EINVAL;
        for (int co = 0; co < cout; ++co) hb[rgbd_cperm(co, perm_out)] = b->v[co];
    }
    HIP_TRY(hipMalloc((void**)&pc->w, n * sizeof(float)));
    if (gen) gen->p.push_back(pc->w);  // registered at once: a later failure leaves nothing behind
    HIP_TRY(hipMalloc((void**)&pc->bias, hb.size() * sizeof(float)));
    if (gen) gen->p.push_back(pc->bias);
    HIP_TRY(hipMemcpy(pc->w, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pc->bias, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice));
    return RGBD_OK;
}

// ConvTranspose2d(cin -> cout <= 4, k = 5, stride 2, pad 2, output_padding 1) as ONE stride-1 3x3 conv over the input grid
// with 16 output channels = 4 output phases x 4: the per-phase form pads the couts to 16 for each of its 25 taps, this one
// runs 9 taps for all phases together (2.8x fewer MFMAs).  Output phase (ry, rx) at input offset (dy, dx) uses kernel
// element ky = ry + 2 - 2 dy, kx = rx + 2 - 2 dx when that is inside the kernel, else a zero weight; the taps run dy, dx =
// 1, 0, -1, which keeps every phase's real taps in the order make_taps() gives them -- with fma(0, x, acc) == acc the
// value of every output is the same chain as in the per-phase form (tests/test_gpu_conv.py::test_subpixel_deconv).
inline int pack_subpix(const HostTensor& w, const HostTensor* b, PackedConv* pc, DevGen* gen, int perm_in = 0)
{
    if (w.shape.size() != 4 || w.shape[2] != 5 || w.shape[3] != 5 || w.shape[1] > 4) return RGBD_EINVAL;
    const int cin = (int)w.shape[0], cout = (int)w.shape[1];
    pc->cin = cin;
    pc->cout = cout;
    pc->k = 5;
    pc->transposed = true;
    pc->subpix = true;
    pc->cin_pad = round_up(cin, 16);
    pc->cout_pad = 16;
    std::vector<float> h((size_t)16 * 9 * pc->cin_pad, 0.f), hb(16, 0.f);
    for (int ry = 0; ry < 2; ++ry)
        for (int rx = 0; rx < 2; ++rx)
            for (int co = 0; co < cout; ++co) {
                const int row = (ry * 2 + rx) * 4 + co;
                if (b) hb[row] = b->v[co];
                for (int u = 0; u < 9; ++u) {
                    const int dy = 1 - u / 3, dx = 1 - u % 3;
                    const int ky = ry + 2 - 2 * dy, kx = rx + 2 - 2 * dx;
                    if (ky < 0 || ky > 4 || kx < 0 || kx > 4) continue;
                    for (int ci = 0; ci < cin; ++ci)
                        h[((size_t)row * 9 + u) * pc->cin_pad + rgbd_cperm(ci, perm_in)] = w.v[(((size_t)ci * cout + co) * 5 + ky) * 5 + kx];
                }
            }
    HIP_TRY(hipMalloc((void**)&pc->w, h.size() * sizeof(float)));
    if (gen) gen->p.push_back(pc->w);
    HIP_TRY(hipMalloc((void**)&pc->bias, hb.size() * sizeof(float)));
    if (gen) gen->p.push_back(pc->bias);
    HIP_TRY(hipMemcpy(pc->w, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pc->bias, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice));
    return RGBD_OK;
}

// Conv2d(cin <= 3 -> cout, k 5, stride 2, pad 2) as a 1x1 layer over the packed input of launch_im2col5s2: weight of
// term n = tap * cin + c at packed index (n / 16) * 16 + (n % 4) * 4 + (n % 16) / 4
inline int pack_kpack(const HostTensor& w, const HostTensor* b, PackedConv* pc, DevGen* gen, int perm_out = 0)
{
    if (w.shape.size() != 4 || w.shape[2] != 5 || w.shape[3] != 5 || w.shape[1] > 3) return RGBD_EINVAL;
    const int cout = (int)w.shape[0], cin = (int)w.shape[1], nterm = 25 * cin;
    HostTensor w1;
    const int KP = round_up(nterm, 16);
    w1.shape = {cout, KP, 1, 1};
    w1.v.assign((size_t)cout * KP, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int n = 0; n < nterm; ++n) {
            const int t = n / cin, c = n % cin, r = n % 16;
            w1.v[(size_t)co * KP + (n / 16) * 16 + (r % 4) * 4 + r / 4] = w.v[((size_t)co * cin + c) * 25 + t];
        }
    const int rc = pack_conv(w1, b, false, pc, gen, 0, perm_out);
    pc->cin = nterm;  // FLOP accounting: the real reduction length
    return rc;
}

inline void make_taps_subpix(ConvArgs* a)
{
    memset(&a->taps, 0, sizeof(a->taps));
    a->nphase = 1;
    a->IS = 1;
    a->OS = 2;
    a->subpix = 1;
    for (int u = 0; u < 9; ++u) {
        a->taps.dy[0][u] = (int8_t)(1 - u / 3);
        a->taps.dx[0][u] = (int8_t)(1 - u % 3);
        a->taps.wt[0][u] = (int8_t)u;
    }
    a->taps.n[0] = 9;
    a->min_dy = a->min_dx = -1;
    a->span_y = a->span_x = 3;
}

inline void make_taps(const PackedConv& pc, int stride, int pad, ConvArgs* a)
{
    const int k = pc.k;
    memset(&a->taps, 0, sizeof(a->taps));
    if (!pc.transposed) {
        a->nphase = 1;
        a->IS = stride;
        a->OS = 1;
        int n = 0;
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx) {
                a->taps.dy[0][n] = (int8_t)(ky - pad);
                a->taps.dx[0][n] = (int8_t)(kx - pad);
                a->taps.wt[0][n] = (int8_t)(ky * k + kx);
                ++n;
            }
        a->taps.n[0] = (int8_t)n;
        a->min_dy = a->min_dx = -pad;
        a->span_y = a->span_x = k;
        return;
    }
    // transposed: o = i*s - pad + k  =>  for o = s*t + r: i = t + (r + pad - k)/s for k == (r + pad) mod s
    a->nphase = stride * stride;
    a->IS = 1;
    a->OS = stride;
    int mn = 127, mx = -127;
    for (int ry = 0; ry < stride; ++ry)
        for (int rx = 0; rx < stride; ++rx) {
            const int ph = ry * stride + rx;
            int n = 0;
            for (int ky = 0; ky < k; ++ky) {
                if ((ry + pad - ky) % stride) continue;
                for (int kx = 0; kx < k; ++kx) {
                    if ((rx + pad - kx) % stride) continue;
                    const int dy = (ry + pad - ky) / stride, dx = (rx + pad - kx) / stride;
                    a->taps.dy[ph][n] = (int8_t)dy;
                    a->taps.dx[ph][n] = (int8_t)dx;
                    a->taps.wt[ph][n] = (int8_t)(ky * k + kx);
                    mn = std::min(mn, std::min(dy, dx));
                    mx = std::max(mx, std::max(dy, dx));
                    ++n;
                }
            }
            a->taps.n[ph] = (int8_t)n;
        }
    a->min_dy = a->min_dx = mn;
    a->span_y = a->span_x = mx - mn + 1;
}


// ---- hang diagnostics (RGBD_DEBUG_DESTROY=1) -----------------------------------------------------------------------------
// A HangWatch around a runtime call that may wait for the device (hipFree = implicit device synchronise): if the call has
// not returned after `secs`, every thread of the process prints its host backtrace (SIGUSR2 handler; the runtime is
// stripped, but the exported entry points -- hipFree, hipGraphLaunch, hsa_signal_wait_*, pthread lock waits -- tell a
// host lock cycle from a wait for a GPU signal), then every engine stream is queried (hipStreamQuery does not block), which
// names the stream that still holds work.  This is how the round-2 "hipFree never returns" report was taken apart.
inline std::mutex g_live_mu;
inline std::map<const void*, hipStream_t> g_live_streams;  // engine -> the stream of its last call
inline const bool g_dbg_destroy = getenv("RGBD_DEBUG_DESTROY") != nullptr;

inline void bt_handler(int)
{
    void* fr[64];
    const int n = backtrace(fr, 64);
    char hdr[64];
    const int l = snprintf(hdr, sizeof(hdr), "[bt tid %ld]\n", (long)syscall(SYS_gettid));
    if (l > 0) (void)!write(2, hdr, (size_t)l);
    backtrace_symbols_fd(fr, n, 2);
}

struct HangWatch {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    HangWatch(const char* what, int secs, bool always = false)
    {
        if (!g_dbg_destroy && !always) return;
        th = std::thread([this, what, secs] {
            std::unique_lock<std::mutex> lk(mu);
            if (cv.wait_for(lk, std::chrono::seconds(secs), [this] { return done; })) return;
            lk.unlock();
            if (!g_dbg_destroy) {
                // production (always-armed) mode is PASSIVE: one line naming the call.  Signalling every thread of the host
                // application, replacing its SIGUSR2 handler and querying streams from here is for RGBD_DEBUG_DESTROY=1 only
                // (round-4 advisor finding): hipFree legitimately waits for the device, a caller's long kernels can be the cause.
                fprintf(stderr, "[rgbd_amd] %s has not returned after %d s (RGBD_DEBUG_DESTROY=1 prints host backtraces)\n", what, secs);
                return;
            }
            fprintf(stderr, "[watchdog] %s has not returned after %d s; host backtraces of every thread follow\n", what, secs);
            void* warm[4];
            (void)backtrace(warm, 4);  // loads libgcc outside the signal handler
            struct sigaction sa, old_sa;
            memset(&sa, 0, sizeof(sa));
            sa.sa_handler = bt_handler;
            sa.sa_flags = SA_RESTART;
            sigaction(SIGUSR2, &sa, &old_sa);
            const long self = (long)syscall(SYS_gettid);
            if (DIR* d = opendir("/proc/self/task")) {
                while (dirent* e = readdir(d)) {
                    const long tid = atol(e->d_name);
                    if (tid <= 0 || tid == self) continue;
                    syscall(SYS_tgkill, (long)getpid(), tid, SIGUSR2);
                    usleep(200 * 1000);
                }
                closedir(d);
            }
            std::map<const void*, hipStream_t> live;
            {
                std::lock_guard<std::mutex> g(g_live_mu);
                live = g_live_streams;
            }
            for (const auto& kv : live) {
                fprintf(stderr, "[watchdog] query stream %p of engine %p ...\n", (void*)kv.second, kv.first);
                fflush(stderr);
                const hipError_t e = hipStreamQuery(kv.second);
                fprintf(stderr, "[watchdog]   -> %s\n", hipGetErrorName(e));
            }
            fprintf(stderr, "[watchdog] query NULL stream ...\n");
            fflush(stderr);
            const hipError_t e0 = hipStreamQuery(nullptr);
            fprintf(stderr, "[watchdog]   -> %s\n", hipGetErrorName(e0));
            fflush(stderr);
            sigaction(SIGUSR2, &old_sa, nullptr);  // the host application's handler is back
            if (getenv("RGBD_DIAG_EXIT")) _exit(86);  // diagnostics runs end by themselves instead of at a time limit
        });
    }
    ~HangWatch()
    {
        if (!th.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            done = true;
        }
        cv.notify_all();
        th.join();
    }
};

inline int g_cfg_epoch = 0;     // bumped by every debug switch that changes kernel choices: cached HIP graphs of older epochs are not reused
inline int g_force_splitk = 0;  // test hook (rgbd_debug_force_splitk)
inline int g_bench_streams = 1;  // rgbd_debug_bench_streams: rgbd_conv_bench issues every launch on this many streams at once
inline const bool g_kpack = !getenv("RGBD_NO_KPACK");  // A/B switch: first analysis conv over a K-packed input (1x1, K = 80 / 32)
inline int g_subpix = getenv("RGBD_NO_SUBPIX") ? 0 : 1;  // rgbd_debug_force_subpix: sub-pixel form of the last transposed conv
inline int g_fail_captures = 0;  // rgbd_debug_fail_captures: the next n graph captures count as lost (test hook)
inline int g_pair = getenv("RGBD_NO_PAIR") ? 0 : 1;  // rgbd_debug_force_pair: RGB / depth layer pairs as one grouped launch
inline int g_force_ckbd = 0;    // test hook (rgbd_debug_force_ckbd): checkerboard output mode of rgbd_conv2d_nchw / rgbd_conv_bench
inline int g_force_blocked = 0;  // rgbd_debug_force_blocked: rgbd_conv_bench launches the blocked-accumulation kernels (tile tuner)
inline const bool g_ckbd_conv = !getenv("RGBD_NO_CKBD_CONV");  // A/B switch: checkerboard-restricted entropy-parameter convs

// ------------------------------------------------------------------------------------------------
// the model
// ------------------------------------------------------------------------------------------------
// Workspace of one engine instance: a two-ended stack.  `top` is the fill of the ACTIVE end (the low end grows up from the
// base, the high end down from base + cap), `other` the fill of the other one.  Blocks allocate and release stack-style on
// the active end (`mark = top ... top = mark`); the stage loops of the big transforms alternate the ends (flip), so that a
// stage's output and temporaries go to the end that holds nothing live any more -- the input of the previous stage -- and
// the workspace holds two consecutive stages instead of the whole transform (round 4: 6.0 -> 2.9 GiB per c3 instance).
// Addresses are a function of the call shape and of cap; a re-allocation (new cap) drops the cached graphs.
struct Arena {
    unsigned char* base = nullptr;
    size_t cap = 0, top = 0, other = 0, peak = 0;
    bool hi = false;
    bool dry = false;
    void* take(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        void* p;
        if (dry) p = (void*)(uintptr_t)(0x1000 + top);
        else p = hi ? (void*)(base + cap - top - bytes) : (void*)(base + top);
        top += bytes;
        peak = std::max(peak, top + other);
        return p;
    }
    void reset()
    {
        top = other = 0;
        hi = false;
    }
    // make the other end the active one, emptied down to `floor` (what is below belongs to somebody who is still alive)
    void flip(size_t floor)
    {
        std::swap(top, other);
        hi = !hi;
        top = floor;
    }
};

struct Epi {
    int act = ACT_NONE;
    const Act* res1 = nullptr;
    const Act* mul = nullptr;
    const Act* res2 = nullptr;
    int ckbd = 0;  // ConvArgs::ckbd: compute / store only one checkerboard half of the output
    const Act* dup = nullptr;  // ConvArgs::y2: the output is also written here (same shape, own channel stride)
};



}  // namespace rgbd_rt
using namespace rgbd_rt;

struct rgbd_elic {
    int N = 192, M = 320;
    int tile_mode = 0;  // rgbd_elic_set_tile_mode: 0 latency tiles (isolated launches), 1 throughput tiles (shared chip)
    int variant = 0;  // 0: ELIC_united (RGB + depth), 1: single-modal ELIC (models/elic.py)
    int in_ch = 3;    // image channels of the single-modal variant
    std::vector<int> slice_ch;
    std::map<std::string, HostTensor> raw;
    std::map<std::string, PackedConv> convs;
    std::map<std::string, float*> dense;  // SE fc weights, EB medians (device)
    TableSet tables[4];
    float* scale_table = nullptr;
    std::shared_ptr<DevGen> gen_w;      // owner of every pointer in convs / dense
    std::shared_ptr<DevGen> gen_scale;  // owner of scale_table
    bool finalized = false;

    Arena arena;
    hipStream_t s = nullptr;
    int rc = 0;
    std::map<std::string, Act> named;  // intermediates of the last call (live in the arena)

    // last compress() results (host)
    std::vector<std::vector<uint8_t>> streams[2][2];
    // last compress() symbol buffers (device, inside the arena) for debug
    int32_t* dbg_sym = nullptr;
    int32_t* dbg_idx = nullptr;
    int64_t dbg_per_mod = 0;
    // rgbd_elic_set_debug_floats: the encoder also keeps, per symbol and in stream order, the value it rounded (y - mean)
    // and the scale it indexed -- what the parity bookkeeping compares with the reference's floats at a flipped symbol
    bool debug_floats = false;
    float* dbg_x = nullptr;
    float* dbg_s = nullptr;

    // rgbd_elic_set_forced_symbols (teacher forcing, parity bookkeeping): the next compress() calls still take every decision
    // from their own floats (symbols / indexes / streams are the GPU's), but what later contexts see is rebuilt from THESE
    // symbols -- z_hat = forced z symbol + median after the z stage, y_hat = forced symbol + mean after every coding part --
    // so that the parts behind a first flip are evaluated under the reference's context (elic_united.py:265-348)
    std::vector<int32_t> force_y[2], force_z[2];

    bool is_clone = false;  // created by rgbd_elic_clone_shared: shares the parent's buffer generations (DevGen)

    // ---- reference arithmetic ----------------------------------------------------------------------
    // refnum: every float operation that feeds a coding decision is performed in the order and with the roundings of the CPU
    // kernels the reference runs on (torch CPU: oneDNN convolutions, Sleef sigmoid, ...): blocked accumulation in the
    // convolutions (conv_mfma_blk.hip), channels stored permuted inside their groups of 16 (rgbd_cperm) so that the MFMA
    // k order is ascending channels.  STF_united keeps the k-ordered single-chain arithmetic of rounds 1-4 (its channel
    // slices are not 16-aligned).  ref_blocks: the reduce blocks of the reference's 1x1 kernels per layer shape, measured on
    // the reference machine (tools/refarith/discover.py -> refarith_tables.json -> rgbd_elic_set_ref_blocks).
    bool refnum = getenv("RGBD_LEGACY_NUMERICS") == nullptr;
    int ref_batch = 1;  // the batch size of the reference call this call stands for (per-image streams: 1)
    int ref_threads = 8;  // CPU threads of the reference run the tables describe (refarith_tables.json "meta"; set_ref_blocks kind 4)
    struct RefTables {
        // kind 0 (1x1 reduce blocks): {0, cin, cout, h, w, batch} -> channels per block
        // kind 1 (small-tensor path, im2col + sgemm): {1, cin, cout, h, w, k * 100 + stride * 10 + pad} -> K-block lengths
        std::map<std::array<int, 6>, std::vector<int>> blocks;
        int misses = 0;
    };
    std::shared_ptr<RefTables> ref_tab = std::make_shared<RefTables>();
    int perm() const { return refnum ? 1 : 0; }
    const std::vector<int>* ref_blocks(int kind, int cin, int cout, int h, int w, int last = -1) const
    {
        auto it = ref_tab->blocks.find({kind, cin, cout, h, w, last < 0 ? ref_batch : last});
        if (it == ref_tab->blocks.end()) {
            ++ref_tab->misses;
            static const bool warn = getenv("RGBD_REFARITH_DEBUG") != nullptr;
            if (warn) fprintf(stderr, "[rgbd_amd] no reference block table for kind %d cin %d cout %d %dx%d batch %d\n", kind, cin, cout, h, w, ref_batch);
            return nullptr;
        }
        return &it->second;
    }

    // conv-kernel profiling (bench.py roofline): HIP event pairs around every conv launch on the launch stream
    bool profile = false;
    bool profile_keys = false;  // rgbd_elic_set_profile(m, 2): the recorded layer names carry the launch's shape key (tools/tune_insitu.py)
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct EvName {
        std::string first;  // layer name
        double second;      // algorithmic FLOPs of the launch (the reference's layer)
        double exec;        // FLOPs the launch executes (checkerboard-output / half-tap launches: less)
    };
    std::vector<EvName> ev_names;  // per recorded launch
    struct LayerAcc {
        double first = 0.0, second = 0.0, exec = 0.0;  // ms, algorithmic FLOPs, executed FLOPs
    };
    std::map<std::string, LayerAcc> prof_layers;
    std::map<std::string, int> prof_counts;
    double prof_flops = 0.0;   // algorithmic (unpadded) FLOPs of the recorded launches: the reference's layers
    double prof_flops_exec = 0.0;  // ... and what the launches execute of them (round-4 review: a checkerboard-output launch computes
                                   // one half of its layer's outputs, an anchor-input launch half of the taps as well)
    double prof_ms = 0.0;
    int64_t prof_launches = 0;

    // --- small helpers -------------------------------------------------------------------------
    bool dry() const { return arena.dry; }

    // ---- HIP graphs ---------------------------------------------------------------------------------------------
    // The launch sequence of a compress() / decompress() call (~750 dependent kernels for ELIC_united) depends only on
    // the call shape: workspace addresses are a deterministic function of (B, H, W, stream format), weights and tables
    // are fixed.  The second call of a shape therefore captures its "body" -- everything between the upload of the
    // inputs and the fetch of the results -- into a HIP graph, and later calls replay it with one hipGraphLaunch: no
    // per-launch host work (name lookups, tap tables, tile choice, argument marshalling), which is what the 16 host
    // threads of a pooled rank used to burn their cores on.  Anything that would stale a baked pointer or a baked
    // kernel choice drops the graphs: workspace re-allocation, new weights / tables, tile-mode and debug switches.
    struct GraphEntry {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        int seen = 0;                       // completed calls of this shape (the first one runs eagerly)
        int capture_fails = 0;              // failed capture attempts; kMaxCaptureFails of them retire the entry to eager launches
        uint64_t last_use = 0;              // graph_clock at the entry's last call (least-recently-used eviction)
        std::map<std::string, Act> named;   // debug tensors of the body (same workspace addresses on every replay)
        Act out[6];                         // body outputs the epilogue reads (x_hat / y_hat per modality; forward(): + likelihoods)
    };
    std::map<std::string, GraphEntry> graphs;
    static constexpr int kMaxCaptureFails = 3;
    static constexpr size_t kMaxGraphs = 24;  // instantiated graphs kept per engine instance (~750 nodes each)
    uint64_t graph_clock = 0;
    bool capture_failed = false;   // the call in progress lost its capture (body_end / a launch inside the capture failed)
    GraphEntry* cur_ge = nullptr;  // entry of the call in progress (nullptr: graphs off for this call)
    int body_mode = 0;             // 0 eager, 1 capturing, 2 replaying
    const bool use_graphs = getenv("RGBD_NO_GRAPH") == nullptr;
    const bool blocking_wait = getenv("RGBD_SPIN_WAIT") == nullptr;
    hipEvent_t done_ev = nullptr;  // blocking-sync event: the host thread sleeps instead of spinning on the stream
    // The legacy NULL stream cannot be captured: a caller that passes it runs the eager launch path (same results, no
    // graph).  Substituting an engine-owned stream for it was built in round 2 and taken out: with it, host waits inside
    // the runtime stopped returning once a pool had switched the device to blocking sync (DESIGN.md 3.5 and
    // profiles/r03_hang_diagnosis.txt have the analysis).  The switch that re-created that configuration is gone from the
    // product (round 4); throughput users drive their own streams (CodecPool), which do capture.
    int use_stream(void* stream)
    {
        s = (hipStream_t)stream;
        if (g_dbg_destroy) {
            std::lock_guard<std::mutex> g(g_live_mu);
            g_live_streams[this] = s;
        }
        return RGBD_OK;
    }
    // pinned staging for the per-call uploads (stream bytes, offsets): truly asynchronous copies, no per-call pinning
    int64_t* res_pin = nullptr;  // pinned landing buffer of the per-call result sizes
    static constexpr size_t kResPinBytes = 64 * 1024;
    void* pin = nullptr;
    size_t pin_cap = 0;
    hipEvent_t pin_ev = nullptr;
    bool pin_busy = false;

    void graphs_invalidate()
    {
        for (auto& kv : graphs) drop_entry(kv.second);
        graphs.clear();
        cur_ge = nullptr;
    }
    GraphEntry* graph_entry(const std::string& key)
    {
        if (!use_graphs || profile || !s) return nullptr;  // (the NULL stream cannot be captured)
        // (the stream is part of the key: a graph is replayed on the stream it was captured on)
        char sk[32];
        snprintf(sk, sizeof(sk), "|%p", (void*)s);
        const std::string cfg = "|" + std::to_string(tile_mode) + "|" + std::to_string(g_cfg_epoch) + "|";
        const std::string full = key + cfg + (sk + 1);
        auto it = graphs.find(full);
        if (it == graphs.end()) {
            // A dataset with many image sizes must not grow this cache without bound: entries of other tile modes / debug
            // epochs can never be replayed again and go first, then the least recently used ones.
            for (auto e = graphs.begin(); e != graphs.end();) {
                // (an entry captured on ANOTHER stream is not stale: an engine used on two streams keeps both sets, the LRU
                //  limit below bounds them)
                const std::string& k = e->first;
                const size_t bar = k.rfind('|');
                const bool stale = bar == std::string::npos || bar + 1 < cfg.size() ||
                                   k.compare(bar + 1 - cfg.size(), cfg.size(), cfg) != 0;
                if (stale) {
                    drop_entry(e->second);
                    e = graphs.erase(e);
                } else {
                    ++e;
                }
            }
            while (graphs.size() >= kMaxGraphs) {
                auto lru = graphs.begin();
                for (auto e = graphs.begin(); e != graphs.end(); ++e)
                    if (e->second.last_use < lru->second.last_use) lru = e;
                drop_entry(lru->second);
                graphs.erase(lru);
            }
            it = graphs.emplace(full, GraphEntry()).first;
        }
        it->second.last_use = ++graph_clock;
        return &it->second;
    }
    static void drop_entry(GraphEntry& ge)
    {
        if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
        if (ge.graph) (void)hipGraphDestroy(ge.graph);
        ge.exec = nullptr;
        ge.graph = nullptr;
    }
    // Start of the capturable part of a call.  Returns true when the caller has to run the body code (eagerly, into a
    // capture, or as a sizing pass), false when a cached graph stands in for it.
    bool body_begin()
    {
        body_mode = 0;
        if (dry() || !cur_ge) return true;
        if (cur_ge->exec) {
            body_mode = 2;
            return false;
        }
        if (cur_ge->seen < 1 || cur_ge->capture_fails >= kMaxCaptureFails) return true;  // first call / retired entry: eager
        // relaxed: other host threads (other engine instances) keep launching while this one captures; the operations
        // that must not overlap a capture are fenced off with g_capture_mu
        g_capture_mu.lock_shared();
        if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) {
            g_capture_mu.unlock_shared();
            (void)hipGetLastError();
            return true;
        }
        body_mode = 1;
        return true;
    }
    int body_end()
    {
        const int mode = body_mode;
        body_mode = 0;
        if (dry()) return RGBD_OK;
        if (mode == 1) {
            hipGraph_t gr = nullptr;
            hipError_t e = hipStreamEndCapture(s, &gr);
            g_capture_mu.unlock_shared();
            if (g_fail_captures > 0) {  // test hook (rgbd_debug_fail_captures): this capture counts as lost
                --g_fail_captures;
                e = hipErrorStreamCaptureInvalidated;
            }
            if (e != hipSuccess || rc) {
                // Nothing of the body has run (it was being recorded, not executed): the caller re-runs the call eagerly
                // (run_sized).  A capture is lost when anything the runtime forbids during a capture happens on this
                // stream's behalf -- another library's device-wide call, an allocator trim -- not through any fault of
                // the call itself.
                if (gr) (void)hipGraphDestroy(gr);
                (void)hipGetLastError();
                capture_failed = true;
                return rc ? rc : RGBD_EHIP;
            }
            hipGraphExec_t ex = nullptr;
            if (hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGraphDestroy(gr);
                (void)hipGetLastError();
                capture_failed = true;
                return RGBD_EHIP;
            }
            cur_ge->graph = gr;
            cur_ge->exec = ex;
            cur_ge->named = named;
        }
        if (mode) {
            HIP_TRY(hipGraphLaunch(cur_ge->exec, s));
            if (mode == 2) named = cur_ge->named;
        }
        if (cur_ge && !rc) ++cur_ge->seen;
        return RGBD_OK;
    }
    // an error return between body_begin() and body_end() must not leave the stream capturing
    void body_abort()
    {
        if (body_mode == 1) {
            capture_failed = true;  // an error inside a capture: the eager re-run tells a lost capture from a real fault
            hipGraph_t gr = nullptr;
            (void)hipStreamEndCapture(s, &gr);
            g_capture_mu.unlock_shared();
            if (gr) (void)hipGraphDestroy(gr);
            (void)hipGetLastError();
        }
        body_mode = 0;
    }
    // Is this engine's stream recording into a graph right now?  (Its own body, or -- when two users were handed the same
    // HIP stream, e.g. past torch's pool of 32 side streams -- somebody else's.)  The NULL stream never captures.
    bool capturing() const
    {
        if (!s) return false;
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return st != hipStreamCaptureStatusNone;
    }
    int wait_stream()
    {
        if (!blocking_wait) {
            HIP_TRY(hipStreamSynchronize(s));
            return RGBD_OK;
        }
        if (!done_ev) HIP_TRY(hipEventCreateWithFlags(&done_ev, hipEventBlockingSync | hipEventDisableTiming));
        if (capturing()) return RGBD_ESTATE;  // an event recorded inside a capture never signals: refuse by construction
        HIP_TRY(hipEventRecord(done_ev, s));
        HangWatch w("hipEventSynchronize(done_ev) in wait_stream", 30);
        HIP_TRY(hipEventSynchronize(done_ev));
        return RGBD_OK;
    }
    // pinned staging buffer of at least `bytes`; waits until the previous call's copies out of it have finished
    int pin_take(size_t bytes, void** out)
    {
        if (pin_busy) {
            HangWatch w("hipEventSynchronize(pin_ev) in pin_take", 30);
            HIP_TRY(hipEventSynchronize(pin_ev));
            pin_busy = false;
        }
        if (bytes > pin_cap) {
            std::unique_lock<std::shared_mutex> lk(g_capture_mu);
            if (pin) (void)hipHostFree(pin);
            pin = nullptr;
            pin_cap = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            HIP_TRY(hipHostMalloc(&pin, want, hipHostMallocDefault));
            pin_cap = want;
        }
        *out = pin;
        return RGBD_OK;
    }
    int pin_release()
    {
        if (!pin_ev) HIP_TRY(hipEventCreateWithFlags(&pin_ev, hipEventBlockingSync | hipEventDisableTiming));
        if (capturing()) return RGBD_ESTATE;  // (see wait_stream)
        HIP_TRY(hipEventRecord(pin_ev, s));
        pin_busy = true;
        return RGBD_OK;
    }
    void fail(int code)
    {
        if (!rc) rc = code;
    }
    Act alloc(int n, int h, int w, int c)
    {
        Act a;
        a.n = n;
        a.h = h;
        a.w = w;
        a.c = c;
        a.cs = round_up(c, 16);
        a.p = (float*)arena.take(a.elems() * sizeof(float));
        return a;
    }
    // ---- two-ended workspace (Arena): the stage loops of g_a / g_s -------------------------------------------------
    // A stage reads tensors on one end and puts its output and temporaries on the other, which is emptied first: what it
    // held -- the previous stage's input -- is dead by then.  `cur_hi`: the end the stage's input lives on.  A fusion stage
    // (its inputs ARE its outputs: the concat buffers) only puts its temporaries there.
    struct Ends {
        size_t lo_floor = 0;  // low-end fill at the transform's entry: everything below belongs to the caller
        bool cur_hi = false;
    };
    Ends ends_begin()
    {
        Ends e;
        if (arena.hi) arena.flip(arena.other);  // (never: transforms are entered on the low end)
        e.lo_floor = arena.top;
        return e;
    }
    void ends_stage(Ends& e, bool output_moves)
    {
        static const bool off = getenv("RGBD_NO_WS_REUSE") != nullptr;  // A/B switch: one-ended workspace as in rounds 1-3
        if (off) return;
        const bool want_hi = !e.cur_hi;
        if (arena.hi != want_hi) arena.flip(want_hi ? 0 : e.lo_floor);
        else arena.top = want_hi ? 0 : e.lo_floor;
        if (output_moves) e.cur_hi = want_hi;
    }
    // back on the low end; the transform's results stay protected on whichever end they are until ends_release()
    void ends_finish(Ends& e)
    {
        if (arena.hi) arena.flip(e.cur_hi ? e.lo_floor : arena.other);
    }
    void ends_release() { arena.other = 0; }

    static Act view(const Act& a, int c0, int c)
    {
        Act v = a;
        v.p = a.p + c0;
        v.c = c;
        return v;
    }
    const PackedConv* conv_of(const std::string& name)
    {
        auto it = convs.find(name);
        if (it == convs.end()) {
            fprintf(stderr, "[rgbd_amd] missing layer %s\n", name.c_str());
            fail(RGBD_ESTATE);
            return nullptr;
        }
        return &it->second;
    }
    // per-row dot-product classes of an SE_Block Linear layer (reference arithmetic; nullptr = every row "main")
    const int* cls_of(const std::string& wname)
    {
        auto it = dense.find(wname + ".rowclass");
        return (it == dense.end() || ref_batch != 1) ? nullptr : reinterpret_cast<const int*>(it->second);
    }
    // the SE Linear layers of a reference call on a batch of two vectors take MKL's n = 2 form for every row (se_linear_ref_kernel
    // form 3; measured for batch 2 only: other batch sizes keep the "main" order and make no bit-level claim)
    int se_form() const { return ref_batch == 2 ? 3 : -1; }
    float* dense_of(const std::string& name)
    {
        auto it = dense.find(name);
        if (it == dense.end()) {
            fprintf(stderr, "[rgbd_amd] missing tensor %s\n", name.c_str());
            fail(RGBD_ESTATE);
            return nullptr;
        }
        return it->second;
    }

    // --- operators -----------------------------------------------------------------------------
    // A convolution is planned (layer lookup, shapes, ConvArgs) and then issued.  Two plans of the same shape on independent
    // data -- the RGB and the depth branch of a transform stage -- are issued as ONE grouped launch (ConvArgs::groups = 2:
    // twice the workgroups, half the launches; every output keeps its fma chain, so the results are those of the two
    // launches bit for bit: tests/test_gpu_pairs.py).
    struct ConvPlan {
        ConvArgs a{};
        Act y;
        std::string name;
        double flops = 0.0;
        double flops_exec = 0.0;  // (see prof_flops_exec)
        size_t partial_bytes = 0;  // split-K / GELU partial planes of this layer
        bool fused = false;        // launch_conv_fused
        bool ok = false;           // a launch is wanted (not a dry run, no error so far)
    };

    // fuse1x1: name of a 1x1 layer applied to relu(conv(x)) inside the same launch (launch_conv_fused); ep / dst / the
    // returned tensor then describe that second layer's output.  Callers ask fusable() first.
    // lead1x1 / lead_dst (only with fuse1x1): a further 1x1 + ReLU applied to that output inside the same launch -- the
    // leading layer of the block that follows -- written to *lead_dst.
    ConvPlan conv_plan(const std::string& name, const Act& x, int stride, int pad, Epi ep = Epi(), const Act* dst = nullptr,
                       const std::string* fuse1x1 = nullptr, const std::string* lead1x1 = nullptr,
                       const Act* lead_dst = nullptr)
    {
        ConvPlan cp;
        cp.name = name;
        const PackedConv* pc = conv_of(name + ".weight");
        if (!pc) return cp;
        const PackedConv* pc2 = fuse1x1 ? conv_of(*fuse1x1 + ".weight") : nullptr;
        if (fuse1x1 && !pc2) return cp;
        const PackedConv* pc3 = (pc2 && lead1x1 && lead_dst) ? conv_of(*lead1x1 + ".weight") : nullptr;
        if (lead1x1 && !pc3) {
            fail(RGBD_EINVAL);
            return cp;
        }
        const int k = pc->k;
        int OH, OW;
        if (!pc->transposed) {
            OH = (x.h + 2 * pad - k) / stride + 1;
            OW = (x.w + 2 * pad - k) / stride + 1;
        } else {
            OH = (x.h - 1) * stride - 2 * pad + k + (stride - 1);
            OW = (x.w - 1) * stride - 2 * pad + k + (stride - 1);
        }
        const PackedConv* pcy = pc2 ? pc2 : pc;  // the layer that produces y
        Act y = dst ? *dst : alloc(x.n, OH, OW, pcy->cout);
        cp.y = y;
        if (round_up(x.c, 16) != pc->cin_pad || y.h != OH || y.w != OW || y.n != x.n || y.c != pcy->cout ||
            (pc2 && (pc2->k != 1 || pc2->cin_pad != pc->cout_pad || pc2->transposed)) ||
            (pc3 && (pc3->k != 1 || pc3->cin_pad != pc2->cout_pad || pc3->transposed || lead_dst->c != pc3->cout ||
                     lead_dst->h != OH || lead_dst->w != OW || lead_dst->n != x.n))) {
            fprintf(stderr, "[rgbd_amd] shape mismatch at %s: x.c=%d cin=%d y=(%d,%d,%d) expect (%d,%d,%d)\n", name.c_str(),
                    x.c, pc->cin, y.h, y.w, y.c, OH, OW, pc->cout);
            fail(RGBD_EINVAL);
            return cp;
        }
        if (rc) return cp;
        // (the sizing pass runs through the same planning -- pointers are placeholders there -- so that it books exactly the
        //  split-K planes the launch will use: a flat "8 planes per layer" used to be most of the workspace, 2 GB per big-map layer)
        ConvArgs& a = cp.a;
        a.x = x.p;
        a.N = x.n;
        a.H = x.h;
        a.W = x.w;
        a.xcs = x.cs;
        a.cin_pad = pc->cin_pad;
        a.w = pc->w;
        a.ntaps_total = k * k;
        a.bias = pc->bias;
        a.y = y.p;
        a.OH = OH;
        a.OW = OW;
        a.ycs = y.cs;
        a.cout_pad = pc->cout_pad;
        // the last transposed conv (N -> 3 / 1): one 9-tap sub-pixel conv instead of four phases of padded couts
        const PackedConv* sp = nullptr;
        if (pc->transposed && g_subpix && !pc2 && stride == 2 && pad == 2 && k == 5 && !ep.res1 && !ep.mul && !ep.res2) {
            auto it = convs.find(name + ".subpix.weight");
            if (it != convs.end()) sp = &it->second;
        }
        if (sp) {
            a.w = sp->w;
            a.bias = sp->bias;
            a.ntaps_total = 9;
            a.cout_pad = 16;
        }
        // a channel slice narrower than its 16-padded width inside a wider buffer (STF_united: 24 of 48): stop at the
        // slice end; a buffer of its own gets its pad channels zeroed as usual
        a.cout_store = (pcy->cout % 16 && y.cs != round_up(pcy->cout, 16)) ? round_up(pcy->cout, 4) : pcy->cout_pad;
        if (pc2) {
            a.w2 = pc2->w;
            a.bias2 = pc2->bias;
            a.cout2_pad = pc2->cout_pad;
            a.act_mid = ACT_RELU;
        }
        if (pc3) {
            a.w3 = pc3->w;
            a.bias3 = pc3->bias;
            a.y3 = lead_dst->p;
            a.y3cs = lead_dst->cs;
            a.cout3_pad = pc3->cout_pad;
        }
        if (sp) {
            make_taps_subpix(&a);
            a.cout_store = 16;
        } else {
            make_taps(*pc, stride, pad, &a);
        }
        a.GH = pc->transposed ? x.h : OH;
        a.GW = pc->transposed ? x.w : OW;
        a.act = ep.act;
        a.ckbd = ep.ckbd;
        if (ep.dup) {
            a.y2 = ep.dup->p;
            a.y2cs = ep.dup->cs;
        }
        a.loaded = tile_mode;
        if (ep.res1) {
            a.res1 = ep.res1->p;
            a.r1cs = ep.res1->cs;
        }
        if (ep.mul) {
            a.mul = ep.mul->p;
            a.mcs = ep.mul->cs;
        }
        if (ep.res2) {
            a.res2 = ep.res2->p;
            a.r2cs = ep.res2->cs;
        }
        // weight-heavy layers on the small latent grid (entropy model, hyper synthesis): split the reduction
        static const char* const kSplitPrefixes[] = {"rgb_entropy_parameters", "depth_entropy_parameters",
                                                     "rgb_channel_context", "depth_channel_context", "rgb_local_context",
                                                     "depth_local_context", "h_s."};
        a.splitk = 1;
        {
            int mt = 1;
            for (int ph = 0; ph < a.nphase; ++ph) mt = std::max(mt, (int)a.taps.n[ph]);
            bool listed = false;
            for (const char* pre : kSplitPrefixes) listed = listed || name.rfind(pre, 0) == 0;
            // a measured entry (csrc/splitk_table.h) applies to any layer of that shape; the rule only to the listed families.
            // Fused tails, the packed image-facing layers and checkerboard-less sub-pixel forms run unsplit.
            const bool splittable = !pc2 && !sp;
            if (g_force_splitk > 0 && listed) a.splitk = g_force_splitk;
            else if (listed) a.splitk = conv_splitk_for(a.cin_pad, a.cout_pad, mt, (long)OH * OW, a.nphase);
            else if (splittable && g_force_splitk >= 0)
                if (const int t = conv_splitk_table(a.cin_pad, a.cout_pad, mt, (long)OH * OW, a.nphase)) a.splitk = t;
        }
        if (refnum) plan_refnum(cp, name, pc, pc2, pc3, sp != nullptr, x, stride, OH, OW);
        // split-K partial planes; a GELU layer (STF_united's MLP) also goes through the reducer, with a single plane
        cp.partial_bytes = (a.splitk > 1 || a.act == ACT_GELU) ? (size_t)a.splitk * x.n * OH * OW * pc->cout_pad * sizeof(float) : 0;
        cp.flops = 2.0 * (double)x.n * OH * OW * (double)pc->cout * pc->cin * k * k /
                       (pc->transposed ? (double)(stride * stride) : 1.0) +
                   (pc2 ? 2.0 * (double)x.n * OH * OW * (double)pc2->cout * pc2->cin : 0.0) +
                   (pc3 ? 2.0 * (double)x.n * OH * OW * (double)pc3->cout * pc3->cin : 0.0);
        cp.flops_exec = a.ckbd ? 0.5 * cp.flops : cp.flops;
        cp.fused = pc2 != nullptr;
        cp.ok = !dry();
        return cp;
    }

    // The accumulation structure of the reference's CPU kernel for this layer (oracle/cpu_arith.c is the C
    // restatement the GPU results are compared with, bit for bit):
    //   conv, k > 1 (oneDNN jit:avx512_core)      a block per 16 input channels; (S_0 + bias) + S_1 + ...
    //   conv, 1x1   (oneDNN jit_1x1:avx512_core)  the layer shape's reduce blocks (ref_blocks); the first chain starts at the bias
    //   conv_transpose2d, stride 1                a block per 16 input channels; bias last
    // Layers with no decision behind them that have a faster special form keep it (the image-producing sub-pixel layer);
    // stride-2 transposed convs: see deconv_s2_ref().
    void plan_refnum(ConvPlan& cp, const std::string& name, const PackedConv* pc, const PackedConv* pc2, const PackedConv* pc3,
                     bool subpix, const Act& x, int stride, int OH, int OW)
    {
        ConvArgs& a = cp.a;
        a.exact_math = 1;
        const bool kpacked = name.size() > 6 && name.compare(name.size() - 6, 6, ".kpack") == 0;
        if (subpix || kpacked || (pc->transposed && stride != 1)) return;  // (single chain, bias in the epilogue)
        if (pc2) a.tail_bias_init = 1;  // the fused 1x1 tails: one reduce block (fusable_ref() has checked), chains start at the bias
        (void)pc3;
        if (pc->k == 1 && !pc->transposed) {
            a.bias_mode = 2;
            const std::vector<int>* bl = ref_blocks(0, pc->cin, pc->cout, x.h, x.w);
            if (!bl || bl->size() <= 1) {
                a.splitk = 1;  // one block: the single-chain kernel with the bias in front
                return;
            }
            const int nb = (int)bl->size();
            // small grids: the blocks as split-K ranges (the ordered reducer adds the block sums); large maps: in the kernel
            const bool split = (long)OH * OW <= 2048 && nb <= 16 && !pc2;
            if (split) {
                a.splitk = nb;
                int pos = 0;
                for (int b = 0; b < nb; ++b) {
                    a.split_c16[b] = (uint16_t)(pos / 16);
                    pos += (*bl)[b];
                }
                a.split_c16[nb] = (uint16_t)((pos + 15) / 16);
            } else {
                a.splitk = 1;
                if (set_blocks_of(&a, bl->data(), nb)) fail(RGBD_EINVAL);
            }
            return;
        }
        a.splitk = 1;
        a.bias_mode = pc->transposed ? 0 : 1;
        if (set_blocks_of(&a, nullptr, 0)) fail(RGBD_EINVAL);
    }
    static int set_blocks_of(ConvArgs* a, const int* blocks, int nblocks)
    {
        memset(a->blk_end, 0, sizeof(a->blk_end));
        const int n16 = a->cin_pad / 16;
        if (n16 > 256) return RGBD_EINVAL;
        if (!blocks || nblocks <= 0) {
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

    // can the two plans share a launch?  Same layer shape, strides and epilogue, operand by operand
    static bool pairable(const ConvPlan& p, const ConvPlan& q)
    {
        if (!p.ok || !q.ok || p.fused != q.fused) return false;
        const ConvArgs &a = p.a, &b = q.a;
        return a.N == b.N && a.H == b.H && a.W == b.W && a.xcs == b.xcs && a.cin_pad == b.cin_pad &&
               a.ntaps_total == b.ntaps_total && a.OH == b.OH && a.OW == b.OW && a.ycs == b.ycs && a.cout_pad == b.cout_pad &&
               a.cout_store == b.cout_store && a.GH == b.GH && a.GW == b.GW && a.IS == b.IS && a.OS == b.OS &&
               a.nphase == b.nphase && a.min_dy == b.min_dy && a.min_dx == b.min_dx && a.span_y == b.span_y &&
               a.span_x == b.span_x && a.act == b.act && !a.res1 == !b.res1 && a.r1cs == b.r1cs && !a.mul == !b.mul &&
               a.mcs == b.mcs && !a.res2 == !b.res2 && a.r2cs == b.r2cs && a.splitk == b.splitk && a.loaded == b.loaded &&
               a.ckbd == b.ckbd && !a.y2 == !b.y2 && a.y2cs == b.y2cs && a.subpix == b.subpix && !a.w2 == !b.w2 &&
               a.cout2_pad == b.cout2_pad && a.act_mid == b.act_mid && !a.w3 == !b.w3 && a.y3cs == b.y3cs &&
               a.cout3_pad == b.cout3_pad && memcmp(&a.taps, &b.taps, sizeof(TapTable)) == 0 && a.blocked == b.blocked &&
               a.bias_mode == b.bias_mode && a.tail_bias_init == b.tail_bias_init && a.exact_math == b.exact_math &&
               memcmp(a.blk_end, b.blk_end, sizeof(a.blk_end)) == 0 && memcmp(a.split_c16, b.split_c16, sizeof(a.split_c16)) == 0;
    }

    // launch one plan, or two plans as one grouped launch (q != nullptr: the caller has checked pairable())
    void conv_issue(ConvPlan& p, ConvPlan* q = nullptr)
    {
        if (dry()) {  // book the split-K scratch of this launch
            const size_t m0 = arena.top;
            (void)arena.take(p.partial_bytes + (q ? q->partial_bytes : 0));
            arena.top = m0;
            return;
        }
        if (rc || !p.ok || (q && !q->ok)) return;
        ConvArgs a = p.a;
        const size_t pmark = arena.top;
        if (p.partial_bytes) a.partial = (float*)arena.take(p.partial_bytes);
        if (q) {
            const ConvArgs& b = q->a;
            a.groups = 2;
            a.g1.x = b.x;
            a.g1.w = b.w;
            a.g1.bias = b.bias;
            a.g1.y = b.y;
            a.g1.res1 = b.res1;
            a.g1.mul = b.mul;
            a.g1.res2 = b.res2;
            a.g1.y2 = b.y2;
            a.g1.w2 = b.w2;
            a.g1.bias2 = b.bias2;
            a.g1.w3 = b.w3;
            a.g1.bias3 = b.bias3;
            a.g1.y3 = b.y3;
            if (q->partial_bytes) a.g1.partial = (float*)arena.take(q->partial_bytes);
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profile) {
            if (ev_used + 2 > ev_pool.size()) {
                for (int i = 0; i < 256; ++i) {
                    hipEvent_t e;
                    if (hipEventCreate(&e) != hipSuccess) {
                        fail(RGBD_EHIP);
                        return;
                    }
                    ev_pool.push_back(e);
                }
            }
            e0 = ev_pool[ev_used++];
            e1 = ev_pool[ev_used++];
            (void)hipEventRecord(e0, s);
        }
        const int r = p.fused ? launch_conv_fused(a, s) : launch_conv(a, s);
        if (profile) {
            (void)hipEventRecord(e1, s);
            const double fl = p.flops + (q ? q->flops : 0.0), fx = p.flops_exec + (q ? q->flops_exec : 0.0);
            prof_flops += fl;
            prof_flops_exec += fx;
            ++prof_launches;
            if (profile_keys) {
                char key[200];
                snprintf(key, sizeof(key), "|%d,%d,%d,%d,%d,%d,%d,%d,%d|%d", a.N * (a.groups == 2 ? 2 : 1), a.H, a.W, a.cin_pad, a.cout_pad,
                         a.ntaps_total, a.nphase > 1 ? a.OS : a.IS, a.nphase + 10 * a.ckbd + (a.blocked ? 100 : 0),
                         std::max(1, std::min(a.splitk, a.cin_pad / 16)), p.fused ? 1 : 0);
                ev_names.push_back({p.name + key, fl, fx});
            } else {
                ev_names.push_back({p.name, fl, fx});
            }
        }
        arena.top = pmark;  // stream order protects the scratch: later kernels of this stream run after the reducer
        if (r) {
            fprintf(stderr, "[rgbd_amd] conv launch failed at %s (%d)\n", p.name.c_str(), r);
            fail(r);
        }
    }

    // first analysis conv (3 / 1 -> N, k 5, stride 2): gather the 25 x C real inputs of every output pixel, then a 1x1 layer
    // with K = 80 / 32 (pack_kpack); returns false when the layer is not of that kind
    bool conv_kpacked(const std::string& name, const Act& x, int stride, int pad, const Epi& ep, const Act* dst, Act* out)
    {
        auto pcw = convs.find(name + ".weight");
        if (pcw == convs.end()) return false;
        const PackedConv* pc = &pcw->second;
        if (!(g_kpack && g_subpix && !pc->transposed && pc->k == 5 && stride == 2 && pad == 2 && x.c <= 3)) return false;
        auto kp = convs.find(name + ".kpack.weight");
        if (kp == convs.end()) return false;
        const int OH = (x.h + 2 * pad - 5) / stride + 1, OW = (x.w + 2 * pad - 5) / stride + 1;
        *out = dst ? *dst : alloc(x.n, OH, OW, pc->cout);
        const size_t mark = arena.top;
        Act xk = alloc(x.n, OH, OW, kp->second.cin_pad);
        if (!dry() && !rc) {
            const int r = launch_im2col5s2(x.p, x.n, x.h, x.w, x.cs, x.c, xk.p, OH, OW, xk.cs, s);
            if (r) fail(r);
        }
        conv(name + ".kpack", xk, 1, 0, ep, out);
        arena.top = mark;
        return true;
    }

    // The reference's small-tensor route (torch ConvParams::use_mkldnn is false: batch 1, kernel <= 3, <= 20480 input
    // elements -> im2col + MKL sgemm): its own accumulation order, k = c -> ky -> kx in K blocks.
    bool small_tensor_layer(const std::string& name, const Act& x) const  // (x: any tensor on the layer's input grid)
    {
        if (!refnum || ref_batch != 1) return false;
        auto it = convs.find(name + ".weight");
        if (it == convs.end()) return false;
        const PackedConv& pc = it->second;
        return !pc.transposed && !pc.subpix && pc.k <= 3 && (long)pc.cin * x.h * x.w <= 20480;
    }
    Act conv_small(const std::string& name, const Act& x, int stride, int pad, const Epi& ep, const Act* dst)
    {
        const PackedConv* pc = conv_of(name + ".weight");
        if (!pc) return Act();
        const int k = pc->k, OH = (x.h + 2 * pad - k) / stride + 1, OW = (x.w + 2 * pad - k) / stride + 1;
        Act y = dst ? *dst : alloc(x.n, OH, OW, pc->cout);
        if (dry() || rc) return y;
        if (y.h != OH || y.w != OW || y.c != pc->cout || (ep.ckbd && stride != 1)) {
            fail(RGBD_EINVAL);
            return y;
        }
        SmallConvArgs a{};
        a.x = x.p;
        a.w = pc->w;
        a.bias = pc->bias;
        a.y = y.p;
        a.N = x.n;
        a.H = x.h;
        a.W = x.w;
        a.xcs = x.cs;
        a.C = pc->cin;
        a.cin_pad = pc->cin_pad;
        a.O = pc->cout;
        a.OH = OH;
        a.OW = OW;
        a.ycs = y.cs;
        a.K = k;
        a.stride = stride;
        a.pad = pad;
        a.act = ep.act;
        a.ckbd = ep.ckbd;
        if (ep.res1) a.res1 = ep.res1->p, a.r1cs = ep.res1->cs;
        if (ep.mul) a.mul = ep.mul->p, a.mcs = ep.mul->cs;
        if (ep.res2) a.res2 = ep.res2->p, a.r2cs = ep.res2->cs;
        if (ep.dup) a.y2 = ep.dup->p, a.y2cs = ep.dup->cs;
        const int Kt = pc->cin * k * k;
        const std::vector<int>* bl = ref_blocks(1, pc->cin, pc->cout, x.h, x.w, k * 100 + stride * 10 + pad);
        a.nb = 1;
        a.kb[0] = 0;
        a.kb[1] = Kt;
        if (bl && bl->size() <= 16) {
            int pos = 0;
            a.nb = (int)bl->size();
            for (int b = 0; b < a.nb; ++b) {
                pos += (*bl)[b];
                a.kb[b + 1] = pos;
            }
            if (pos != Kt) {
                fail(RGBD_EINVAL);
                return y;
            }
        }
        const int r = launch_small_conv_ref(a, s);
        if (r) fail(r);
        return y;
    }

    // torch.sigmoid on the CPU is not one function: the last len % 32 elements of every parallel chunk of the
    // tensor go through the scalar path (libm's expf instead of Sleef's vector exp).  A tensor that has such elements -- e.g.
    // the 320 x 16 x 16 attention map of a 256 x 256 image: three chunks of 27307 -- cannot take its gate in the conv epilogue,
    // which knows no flat index: the conv then writes its plain output and sigmoid_gate_ref_kernel applies a * sigmoid(b) + x.
    bool sigmoid_scalar_tails(long numel) const
    {
        long tasks = 1;
        if (numel >= 32768 && ref_threads > 1) tasks = std::min<long>(ref_threads, (numel + 32767) / 32768);
        const long chunk = (numel + tasks - 1) / tasks;
        return chunk % 32 != 0 || (numel - (tasks - 1) * chunk) % 32 != 0;
    }
    bool gate_needs_own_pass(const std::string& name, const Act& x, int stride, int pad, const Epi& ep)
    {
        if (!refnum || ep.act != ACT_SIGMOID) return false;
        auto it = convs.find(name + ".weight");
        if (it == convs.end() || it->second.transposed) return false;
        const PackedConv& pc = it->second;
        const int OH = (x.h + 2 * pad - pc.k) / stride + 1, OW = (x.w + 2 * pad - pc.k) / stride + 1;
        return sigmoid_scalar_tails((long)(ref_batch == 1 ? 1 : x.n) * pc.cout * OH * OW);
    }
    Act conv_gated_ref(const std::string& name, const Act& x, int stride, int pad, const Epi& ep, const Act* dst)
    {
        const PackedConv& pc = convs.find(name + ".weight")->second;
        const int OH = (x.h + 2 * pad - pc.k) / stride + 1, OW = (x.w + 2 * pad - pc.k) / stride + 1;
        Act out = dst ? *dst : alloc(x.n, OH, OW, pc.cout);
        const size_t mark = arena.top;
        Epi plain;
        plain.res1 = ep.res1;
        const Act t = conv(name, x, stride, pad, plain);
        if (!dry() && !rc) {
            const int r = launch_sigmoid_gate_ref(t.p, t.cs, ep.mul ? ep.mul->p : nullptr, ep.mul ? ep.mul->cs : 0,
                                                  ep.res2 ? ep.res2->p : nullptr, ep.res2 ? ep.res2->cs : 0, out.p, out.cs, x.n,
                                                  OH * OW, pc.cout, ref_batch == 1 ? 1 : 0, ref_threads, s);
            if (r) fail(r);
        }
        arena.top = mark;
        return out;
    }

    Act conv(const std::string& name, const Act& x, int stride, int pad, Epi ep = Epi(), const Act* dst = nullptr,
             const std::string* fuse1x1 = nullptr, const std::string* lead1x1 = nullptr, const Act* lead_dst = nullptr)
    {
        Act out;
        if (!fuse1x1 && !ep.dup && !ep.ckbd && gate_needs_own_pass(name, x, stride, pad, ep)) return conv_gated_ref(name, x, stride, pad, ep, dst);
        if (!fuse1x1 && conv_kpacked(name, x, stride, pad, ep, dst, &out)) return out;
        if (!fuse1x1 && small_tensor_layer(name, x)) return conv_small(name, x, stride, pad, ep, dst);
        if (refnum && !fuse1x1 && !dst && stride == 2 && pad == 2 && !ep.res1 && !ep.mul && !ep.res2 && !ep.dup && !ep.ckbd) {
            Act o;  // a stride-2 transposed conv with a measured recipe (the hyper-synthesis stages)
            if (deconv_s2_ref(name, x, ep.act, &o)) return o;
        }
        ConvPlan cp = conv_plan(name, x, stride, pad, ep, dst, fuse1x1, lead1x1, lead_dst);
        conv_issue(cp);
        return cp.y;
    }

    // A stride-1 k x k conv whose INPUT is non-zero at the anchor positions only ((row + col) odd: the slice right after its
    // anchor pass, utils/ckbd.py:37-48 -- what the local-context convs read, elic_united.py:296,309).  An output pixel of
    // parity q then only meets non-zero inputs under the taps with (dy + dx) & 1 == 1 - q: the anchor outputs need the 13
    // taps with dy + dx even, the other outputs the 12 with dy + dx odd.  Two checkerboard-output launches (ConvArgs::ckbd
    // 1 / 2), each with its half of the tap table: half the MFMA work, and every output keeps its fma chain minus terms
    // that are exact zeros (round 4; RGBD_NO_ANCHOR_TAPS=1 runs the full conv).  Worth it for the 192-channel slice only
    // (368 -> 255 us at c3); at 64 channels two launches cost more than the taps they save (71 -> 86 us).
    void conv_anchor_in(const std::string& name, const Act& x, int pad, const Act& dst)
    {
        static const bool off = getenv("RGBD_NO_ANCHOR_TAPS") != nullptr;
        ConvPlan cp = conv_plan(name, x, 1, pad, Epi(), &dst);
        if (off || g_force_ckbd || x.c < 128 || (!dry() && (!cp.ok || cp.a.nphase != 1 || cp.a.IS != 1 || cp.a.subpix))) {
            conv_issue(cp);
            return;
        }
        for (int par = 1; par <= 2; ++par) {
            ConvPlan h = cp;
            if (!dry()) {
                TapTable& t = h.a.taps;
                int n = 0;
                for (int k = 0; k < cp.a.taps.n[0]; ++k) {
                    const int odd = (cp.a.taps.dy[0][k] + cp.a.taps.dx[0][k]) & 1;
                    if (odd != (par == 1 ? 0 : 1)) continue;  // anchor outputs (parity 1): dy + dx even
                    t.dy[0][n] = cp.a.taps.dy[0][k];
                    t.dx[0][n] = cp.a.taps.dx[0][k];
                    t.wt[0][n] = cp.a.taps.wt[0][k];
                    ++n;
                }
                for (int k = n; k < 25; ++k) t.dy[0][k] = t.dx[0][k] = t.wt[0][k] = 0;
                t.n[0] = (int8_t)n;
                h.a.ckbd = par;
                h.flops = cp.flops * 0.5;
                h.flops_exec = cp.flops * 0.5 * n / std::max(1, (int)cp.a.taps.n[0]);  // half the outputs, n of the taps
            }
            conv_issue(h);
        }
    }

    // The same layer kind for both modalities (names n[0] / n[1]: RGB / depth branch): one grouped launch when the two
    // plans agree in every shape, otherwise (first / last image-facing layers: 3 vs 1 channels) two launches.
    void conv2(const std::string n[2], const Act x[2], int stride, int pad, const Epi ep[2], const Act* const dst[2], Act out[2],
               const std::string* const fuse1x1[2] = nullptr, const std::string* const lead1x1[2] = nullptr,
               const Act* const lead_dst[2] = nullptr)
    {
        if (!fuse1x1 && (gate_needs_own_pass(n[0], x[0], stride, pad, ep[0]) || gate_needs_own_pass(n[1], x[1], stride, pad, ep[1]))) {
            for (int m = 0; m < 2; ++m) out[m] = conv(n[m], x[m], stride, pad, ep[m], dst ? dst[m] : nullptr);
            return;
        }
        if (!fuse1x1) {
            Act o0;
            if (conv_kpacked(n[0], x[0], stride, pad, ep[0], dst ? dst[0] : nullptr, &o0)) {  // (the depth twin is of that kind too)
                out[0] = o0;
                out[1] = conv(n[1], x[1], stride, pad, ep[1], dst ? dst[1] : nullptr);
                return;
            }
        }
        if (!fuse1x1 && small_tensor_layer(n[0], x[0]) && small_tensor_layer(n[1], x[1])) {
            for (int m = 0; m < 2; ++m) out[m] = conv_small(n[m], x[m], stride, pad, ep[m], dst ? dst[m] : nullptr);
            return;
        }
        ConvPlan p[2];
        for (int m = 0; m < 2; ++m)
            p[m] = conv_plan(n[m], x[m], stride, pad, ep[m], dst ? dst[m] : nullptr, fuse1x1 ? fuse1x1[m] : nullptr,
                             lead1x1 ? lead1x1[m] : nullptr, lead_dst ? lead_dst[m] : nullptr);
        out[0] = p[0].y;
        out[1] = p[1].y;
        if (g_pair && (dry() || pairable(p[0], p[1]))) {
            conv_issue(p[0], &p[1]);
        } else {
            conv_issue(p[0]);
            conv_issue(p[1]);
        }
    }

    // drain recorded event pairs into prof_ms (call after the stream has been synchronised)
    void profile_collect()
    {
        for (size_t i = 0; i + 1 < ev_used; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev_pool[i], ev_pool[i + 1]) == hipSuccess) {
                prof_ms += ms;
                if (i / 2 < ev_names.size()) {
                    auto& acc = prof_layers[ev_names[i / 2].first];
                    acc.first += ms;
                    acc.second += ev_names[i / 2].second;
                    acc.exec += ev_names[i / 2].exec;
                    ++prof_counts[ev_names[i / 2].first];
                }
            }
        }
        ev_used = 0;
        ev_names.clear();
    }

    void copy_ch(const Act& src, const Act& dst)
    {
        if (dry() || rc) return;
        const int r = launch_copy_channels(src.p, src.cs, dst.p, dst.cs, src.n * src.h * src.w, round_up(src.c, 4), s);
        if (r) fail(r);
    }

    // the pair (mid: 3x3 + ReLU, last: 1x1 + residual) as one launch?  (a speed decision: the results are bit-identical)
    bool fusable(const std::string& mid, const std::string& last, const Act& x, int groups = 1)
    {
        auto a = convs.find(mid + ".weight"), b = convs.find(last + ".weight");
        if (a == convs.end() || b == convs.end()) return false;
        const PackedConv &p3 = a->second, &p1 = b->second;
        if (p3.transposed || p1.transposed || p1.k != 1 || p3.k != 3 || p1.cin_pad != p3.cout_pad || p1.cout % 16) return false;
        if (refnum) {  // the fused tail runs its 1x1 as one chain: only when the reference's kernel has one reduce block there
            if (small_tensor_layer(mid, x) || small_tensor_layer(last, x)) return false;
            const std::vector<int>* bl = ref_blocks(0, p1.cin, p1.cout, x.h, x.w);
            if (bl && bl->size() > 1) return false;
        }
        return conv_fused_plan(p3.cout_pad, p1.cout_pad, p3.k * p3.k, x.n * groups, x.h, x.w, tile_mode) > 0;
    }

    // ... and can the leading 1x1 + ReLU of the block after it ride along?  (its input is this block's output)
    bool lead_fusable(const std::string& last, const std::string& lead, const Act& x)
    {
        static const bool off = getenv("RGBD_NO_FUSE_LEAD") != nullptr;
        if (off || g_fuse_lead_off || lead.empty()) return false;
        auto b = convs.find(last + ".weight"), c = convs.find(lead + ".weight");
        if (b == convs.end() || c == convs.end()) return false;
        const PackedConv &p1 = b->second, &p0 = c->second;
        if (refnum) {  // (as in fusable(): the leading 1x1 rides along only as a single reduce block)
            if (small_tensor_layer(lead, x)) return false;
            const std::vector<int>* bl = ref_blocks(0, p0.cin, p0.cout, x.h, x.w);
            if (bl && bl->size() > 1) return false;
        }
        return !p0.transposed && p0.k == 1 && p0.cin_pad == p1.cout_pad && p0.cout_pad == p1.cin_pad && p0.cout % 16 == 0 &&
               p1.cout_pad % 32 == 0;
    }
    // outputs of leading layers that a previous block's launch has already produced, by layer name
    std::map<std::string, Act> pre_leads;
    Act take_lead(const std::string& name, const Act& x)
    {
        auto it = pre_leads.find(name);
        if (it != pre_leads.end()) {
            Act t = it->second;
            pre_leads.erase(it);
            return t;
        }
        Epi relu;
        relu.act = ACT_RELU;
        return conv(name, x, 1, 0, relu);
    }

    // modules/layers/res_blk.py:7-27.  next_lead: the leading layer of the block that consumes this block's output ("" = none)
    Act bottleneck(const std::string& p, const Act& x, const Act* dst = nullptr, const std::string& next_lead = std::string())
    {
        const PackedConv* last = conv_of(p + ".branch.4.weight");
        if (!last) return Act();
        Act out = dst ? *dst : alloc(x.n, x.h, x.w, last->cout);
        const std::string last_name = p + ".branch.4";
        const bool fuse = fusable(p + ".branch.2", last_name, x);
        const bool lead = fuse && lead_fusable(last_name, next_lead, x);
        Act lead_out;
        if (lead) lead_out = alloc(x.n, x.h, x.w, convs.find(next_lead + ".weight")->second.cout);  // outlives this block
        const size_t mark = arena.top;
        Epi relu;
        relu.act = ACT_RELU;
        Act t1 = take_lead(p + ".branch.0", x);
        Epi e;
        Act idn = x;
        if (convs.count(p + ".skip.weight")) idn = conv(p + ".skip", x, 1, 0, Epi(), &out);  // (in place: see bottleneck2)
        e.res1 = &idn;
        if (lead) {
            conv(p + ".branch.2", t1, 1, 1, e, &out, &last_name, &next_lead, &lead_out);
            pre_leads[next_lead] = lead_out;
        } else if (fuse) {
            conv(p + ".branch.2", t1, 1, 1, e, &out, &last_name);
        } else {
            Act t2 = conv(p + ".branch.2", t1, 1, 1, relu);
            conv(last_name, t2, 1, 0, e, &out);
        }
        arena.top = mark;
        return out;
    }

    // CompressAI/compressai/layers/layers.py:177-196
    Act res_unit(const std::string& p, const Act& x, const std::string& next_lead = std::string())
    {
        Act out = alloc(x.n, x.h, x.w, x.c);
        const std::string last_name = p + ".conv.4";
        const bool fuse = fusable(p + ".conv.2", last_name, x);
        const bool lead = fuse && lead_fusable(last_name, next_lead, x);
        Act lead_out;
        if (lead) lead_out = alloc(x.n, x.h, x.w, convs.find(next_lead + ".weight")->second.cout);
        const size_t mark = arena.top;
        Epi relu;
        relu.act = ACT_RELU;
        Act t1 = take_lead(p + ".conv.0", x);
        Epi e;
        e.act = ACT_RELU;
        e.res1 = &x;
        if (lead) {
            conv(p + ".conv.2", t1, 1, 1, e, &out, &last_name, &next_lead, &lead_out);
            pre_leads[next_lead] = lead_out;
        } else if (fuse) {
            conv(p + ".conv.2", t1, 1, 1, e, &out, &last_name);
        } else {
            Act t2 = conv(p + ".conv.2", t1, 1, 1, relu);
            conv(last_name, t2, 1, 0, e, &out);
        }
        arena.top = mark;
        return out;
    }

    // layers.py:198-213
    Act attention(const std::string& p, const Act& x, const Act* dst = nullptr)
    {
        Act out = dst ? *dst : alloc(x.n, x.h, x.w, x.c);
        const size_t mark = arena.top;
        Act a = x;
        for (int u = 0; u < 3; ++u)
            a = res_unit(p + ".conv_a." + std::to_string(u), a,
                         u < 2 ? p + ".conv_a." + std::to_string(u + 1) + ".conv.0" : std::string());
        Act b = x;
        for (int u = 0; u < 3; ++u)
            b = res_unit(p + ".conv_b." + std::to_string(u), b,
                         u < 2 ? p + ".conv_b." + std::to_string(u + 1) + ".conv.0" : std::string());
        Epi e;
        e.act = ACT_SIGMOID;
        e.mul = &a;
        e.res2 = &x;
        conv(p + ".conv_b.3", b, 1, 0, e, &out);
        arena.top = mark;
        return out;
    }

    // ---- the same blocks for both modalities at once (p[0] / p[1]: the RGB / depth branch's layer names) ----------------
    // Every layer pair is one grouped launch (conv2).  The fusion decisions are taken for the pair: a grouped launch tiles
    // like the layer at twice the batch.
    void take_lead2(const std::string n[2], const Act x[2], Act t[2])
    {
        auto i0 = pre_leads.find(n[0]), i1 = pre_leads.find(n[1]);
        if (i0 != pre_leads.end() && i1 != pre_leads.end()) {
            t[0] = i0->second;
            t[1] = i1->second;
            pre_leads.erase(n[0]);
            pre_leads.erase(n[1]);
            return;
        }
        if (i0 != pre_leads.end() || i1 != pre_leads.end()) {  // (never planned that way; stay correct)
            t[0] = take_lead(n[0], x[0]);
            t[1] = take_lead(n[1], x[1]);
            return;
        }
        Epi relu[2];
        relu[0].act = relu[1].act = ACT_RELU;
        conv2(n, x, 1, 0, relu, nullptr, t);
    }

    // res_blk.py:7-27 for both modalities
    void bottleneck2(const std::string p[2], const Act x[2], const Act* const dst[2], const std::string next_lead[2], Act out[2])
    {
        const PackedConv* last[2] = {conv_of(p[0] + ".branch.4.weight"), conv_of(p[1] + ".branch.4.weight")};
        if (!last[0] || !last[1]) return;
        const int G = g_pair ? 2 : 1;
        std::string last_name[2], mid[2], lead0[2];
        bool fuse = true, lead = true;
        const bool skip = convs.count(p[0] + ".skip.weight") && convs.count(p[1] + ".skip.weight");
        for (int m = 0; m < 2; ++m) {
            out[m] = (dst && dst[m]) ? *dst[m] : alloc(x[m].n, x[m].h, x[m].w, last[m]->cout);
            last_name[m] = p[m] + ".branch.4";
            mid[m] = p[m] + ".branch.2";
            lead0[m] = p[m] + ".branch.0";
            // the pair is planned at 2N; should the two plans not share a launch after all (conv2 falls back to two launches
            // when pairable() fails), each of them is re-planned at N -- so fusing has to be possible at both sizes
            fuse = fuse && fusable(mid[m], last_name[m], x[m], G) && fusable(mid[m], last_name[m], x[m], 1);
        }
        for (int m = 0; m < 2; ++m) lead = lead && fuse && lead_fusable(last_name[m], next_lead[m], x[m]);
        Act lead_out[2];
        if (lead)
            for (int m = 0; m < 2; ++m)
                lead_out[m] = alloc(x[m].n, x[m].h, x[m].w, convs.find(next_lead[m] + ".weight")->second.cout);  // outlives this block
        const size_t mark = arena.top;
        Act t1[2];
        take_lead2(lead0, x, t1);
        Act idn[2] = {x[0], x[1]};
        if (skip) {
            // the skip path lands in the block's output buffer and the last layer adds to it in place (each element is read
            // and written by the one thread that owns it): no 2 x 252 MB identity tensor at the workspace's peak stage
            const std::string sk[2] = {p[0] + ".skip", p[1] + ".skip"};
            const Epi none[2];
            const Act* sdst[2] = {&out[0], &out[1]};
            conv2(sk, x, 1, 0, none, sdst, idn);
        } else if (convs.count(p[0] + ".skip.weight") || convs.count(p[1] + ".skip.weight")) {
            fail(RGBD_EINVAL);  // (the two branches are built alike)
            return;
        }
        Epi e[2];
        e[0].res1 = &idn[0];
        e[1].res1 = &idn[1];
        const Act* odst[2] = {&out[0], &out[1]};
        Act o[2];
        if (fuse) {
            const std::string* f1[2] = {&last_name[0], &last_name[1]};
            const std::string* l1[2] = {&next_lead[0], &next_lead[1]};
            const Act* ld[2] = {&lead_out[0], &lead_out[1]};
            conv2(mid, t1, 1, 1, e, odst, o, f1, lead ? l1 : nullptr, lead ? ld : nullptr);
            if (lead)
                for (int m = 0; m < 2; ++m) pre_leads[next_lead[m]] = lead_out[m];
        } else {
            Epi relu[2];
            relu[0].act = relu[1].act = ACT_RELU;
            Act t2[2];
            conv2(mid, t1, 1, 1, relu, nullptr, t2);
            conv2(last_name, t2, 1, 0, e, odst, o);
        }
        arena.top = mark;
    }

    // layers.py:177-196 for both modalities
    void res_unit2(const std::string p[2], const Act x[2], const std::string next_lead[2], Act out[2])
    {
        const int G = g_pair ? 2 : 1;
        std::string last_name[2], mid[2], lead0[2];
        bool fuse = true, lead = true;
        for (int m = 0; m < 2; ++m) {
            out[m] = alloc(x[m].n, x[m].h, x[m].w, x[m].c);
            last_name[m] = p[m] + ".conv.4";
            mid[m] = p[m] + ".conv.2";
            lead0[m] = p[m] + ".conv.0";
            fuse = fuse && fusable(mid[m], last_name[m], x[m], G) && fusable(mid[m], last_name[m], x[m], 1);  // (see bottleneck2)
        }
        for (int m = 0; m < 2; ++m) lead = lead && fuse && lead_fusable(last_name[m], next_lead[m], x[m]);
        Act lead_out[2];
        if (lead)
            for (int m = 0; m < 2; ++m)
                lead_out[m] = alloc(x[m].n, x[m].h, x[m].w, convs.find(next_lead[m] + ".weight")->second.cout);
        const size_t mark = arena.top;
        Act t1[2];
        take_lead2(lead0, x, t1);
        Epi e[2];
        for (int m = 0; m < 2; ++m) {
            e[m].act = ACT_RELU;
            e[m].res1 = &x[m];
        }
        const Act* odst[2] = {&out[0], &out[1]};
        Act o[2];
        if (fuse) {
            const std::string* f1[2] = {&last_name[0], &last_name[1]};
            const std::string* l1[2] = {&next_lead[0], &next_lead[1]};
            const Act* ld[2] = {&lead_out[0], &lead_out[1]};
            conv2(mid, t1, 1, 1, e, odst, o, f1, lead ? l1 : nullptr, lead ? ld : nullptr);
            if (lead)
                for (int m = 0; m < 2; ++m) pre_leads[next_lead[m]] = lead_out[m];
        } else {
            Epi relu[2];
            relu[0].act = relu[1].act = ACT_RELU;
            Act t2[2];
            conv2(mid, t1, 1, 1, relu, nullptr, t2);
            conv2(last_name, t2, 1, 0, e, odst, o);
        }
        arena.top = mark;
    }

    // layers.py:198-213 for both modalities
    void attention2(const std::string p[2], const Act x[2], const Act* const dst[2], Act out[2])
    {
        for (int m = 0; m < 2; ++m) out[m] = (dst && dst[m]) ? *dst[m] : alloc(x[m].n, x[m].h, x[m].w, x[m].c);
        const size_t mark = arena.top;
        Act a[2] = {x[0], x[1]}, b[2] = {x[0], x[1]};
        for (int br = 0; br < 2; ++br) {
            const char* tag = br ? ".conv_b." : ".conv_a.";
            Act* cur = br ? b : a;
            for (int u = 0; u < 3; ++u) {
                std::string n[2], nl[2];
                for (int m = 0; m < 2; ++m) {
                    n[m] = p[m] + tag + std::to_string(u);
                    nl[m] = u < 2 ? p[m] + tag + std::to_string(u + 1) + ".conv.0" : std::string();
                }
                Act o[2];
                res_unit2(n, cur, nl, o);
                cur[0] = o[0];
                cur[1] = o[1];
            }
        }
        Epi e[2];
        for (int m = 0; m < 2; ++m) {
            e[m].act = ACT_SIGMOID;
            e[m].mul = &a[m];
            e[m].res2 = &x[m];
        }
        const std::string n[2] = {p[0] + ".conv_b.3", p[1] + ".conv_b.3"};
        const Act* odst[2] = {&out[0], &out[1]};
        Act o[2];
        conv2(n, b, 1, 0, e, odst, o);
        arena.top = mark;
    }

    // attention.py:84-97 for both modalities: x[m] -> dst[m] = x[m] * sigmoid(...) (+ add[m])
    void esa2(const std::string p[2], const Act x[2], const Act dst[2], const Act* const add[2])
    {
        const size_t mark = arena.top;
        const Epi none[2];
        Epi relu[2];
        relu[0].act = relu[1].act = ACT_RELU;
        auto names = [&](const char* suf, std::string n[2]) {
            n[0] = p[0] + suf;
            n[1] = p[1] + suf;
        };
        std::string n[2];
        Act c1_[2], c1[2];
        names(".conv1", n);
        conv2(n, x, 1, 0, none, nullptr, c1_);
        names(".conv2", n);
        conv2(n, c1_, 2, 0, none, nullptr, c1);
        if (c1[0].h < 7 || c1[0].w < 7) {
            fail(RGBD_EINVAL);
            return;
        }
        const int ph = (c1[0].h - 7) / 3 + 1, pw = (c1[0].w - 7) / 3 + 1;
        Act v[2];
        for (int m = 0; m < 2; ++m) v[m] = alloc(x[m].n, ph, pw, c1[m].c);
        const bool same = g_pair && c1[0].n == c1[1].n && c1[0].h == c1[1].h && c1[0].w == c1[1].w && c1[0].cs == c1[1].cs;
        if (!dry() && !rc) {  // both modalities' pooled branches in one launch when they have the same shape (they do)
            int r = launch_maxpool7s3(c1[0].p, c1[0].n, c1[0].h, c1[0].w, c1[0].cs, v[0].p, ph, pw, s, same ? c1[1].p : nullptr,
                                      same ? v[1].p : nullptr);
            if (!r && !same) r = launch_maxpool7s3(c1[1].p, c1[1].n, c1[1].h, c1[1].w, c1[1].cs, v[1].p, ph, pw, s);
            if (r) fail(r);
        }
        Act vr[2], c3[2], c3b[2], up[2], sum[2], o[2];
        names(".conv_max", n);
        conv2(n, v, 1, 1, relu, nullptr, vr);
        names(".conv3", n);
        conv2(n, vr, 1, 1, relu, nullptr, c3);
        names(".conv3_", n);
        conv2(n, c3, 1, 1, none, nullptr, c3b);
        for (int m = 0; m < 2; ++m) up[m] = alloc(x[m].n, x[m].h, x[m].w, c3b[m].c);
        if (!dry() && !rc) {
            const bool same2 = same && x[0].h == x[1].h && x[0].w == x[1].w && c3b[0].cs == c3b[1].cs && up[0].cs == up[1].cs;
            const int rc0 = refnum ? c3b[0].c : 0, rc1 = refnum ? c3b[1].c : 0;
            int r = launch_bilinear(c3b[0].p, c3b[0].n, c3b[0].h, c3b[0].w, c3b[0].cs, up[0].p, x[0].h, x[0].w, s,
                                    same2 ? c3b[1].p : nullptr, same2 ? up[1].p : nullptr, rc0);
            if (!r && !same2)
                r = launch_bilinear(c3b[1].p, c3b[1].n, c3b[1].h, c3b[1].w, c3b[1].cs, up[1].p, x[1].h, x[1].w, s, nullptr, nullptr, rc1);
            if (r) fail(r);
        }
        Epi addup[2];
        addup[0].res1 = &up[0];
        addup[1].res1 = &up[1];
        names(".conv_f", n);
        conv2(n, c1_, 1, 0, addup, nullptr, sum);
        Epi gate[2];
        for (int m = 0; m < 2; ++m) {
            gate[m].act = ACT_SIGMOID;
            gate[m].mul = &x[m];
            gate[m].res2 = add ? add[m] : nullptr;
        }
        const Act* odst[2] = {&dst[0], &dst[1]};
        names(".conv4", n);
        conv2(n, sum, 1, 0, gate, odst, o);
        arena.top = mark;
    }

    // modules/transform/attention.py:84-97; x: [.., n_feats]; writes x * sigmoid(...) into dst
    void esa(const std::string& p, const Act& x, const Act& dst, const Act* add = nullptr)
    {
        const size_t mark = arena.top;
        Act c1_ = conv(p + ".conv1", x, 1, 0);
        Act c1 = conv(p + ".conv2", c1_, 2, 0);
        const int ph = (c1.h - 7) / 3 + 1, pw = (c1.w - 7) / 3 + 1;
        if (c1.h < 7 || c1.w < 7) {
            fail(RGBD_EINVAL);
            return;
        }
        Act v = alloc(x.n, ph, pw, c1.c);
        if (!dry() && !rc) {
            const int r = launch_maxpool7s3(c1.p, c1.n, c1.h, c1.w, c1.cs, v.p, ph, pw, s);
            if (r) fail(r);
        }
        Epi relu;
        relu.act = ACT_RELU;
        Act vr = conv(p + ".conv_max", v, 1, 1, relu);
        Act c3 = conv(p + ".conv3", vr, 1, 1, relu);
        c3 = conv(p + ".conv3_", c3, 1, 1);
        Act up = alloc(x.n, x.h, x.w, c3.c);
        if (!dry() && !rc) {
            const int r = launch_bilinear(c3.p, c3.n, c3.h, c3.w, c3.cs, up.p, x.h, x.w, s, nullptr, nullptr, refnum ? c3.c : 0);
            if (r) fail(r);
        }
        Epi addup;
        addup.res1 = &up;
        Act sum = conv(p + ".conv_f", c1_, 1, 0, addup);
        Epi gate;
        gate.act = ACT_SIGMOID;
        gate.mul = &x;
        gate.res2 = add;  // STF_united adds the gated features to the stream instead of concatenating them
        conv(p + ".conv4", sum, 1, 0, gate, &dst);
        arena.top = mark;
    }

    // modules/transform/attention.py:35-48; rgb/depth: views of N channels; writes the gated features into
    // r_dst / d_dst (N channels each)
    void bi_spf(const std::string& p, const Act& rgb, const Act& depth, const Act& r_dst, const Act& d_dst,
                bool residual = false)
    {
        const size_t mark = arena.top;
        const int half = rgb.c / 2;
        Act rd = alloc(rgb.n, rgb.h, rgb.w, rgb.c);  // cat(rf, df)
        Act dr = alloc(rgb.n, rgb.h, rgb.w, rgb.c);  // cat(df, rf)
        Epi relu;
        relu.act = ACT_RELU;
        Act rf = view(rd, 0, half), df = view(rd, half, half);
        // each extractor writes its features into both concat buffers (cat(rf, df) and cat(df, rf)) from its epilogue
        const Act rf2 = view(dr, half, half), df2 = view(dr, 0, half);
        Epi er = relu, ed = relu;
        er.dup = &rf2;
        ed.dup = &df2;
        {
            const std::string n[2] = {p + ".r_ext", p + ".d_ext"};
            const Act x[2] = {rgb, depth};
            const Epi ep[2] = {er, ed};
            const Act* dst[2] = {&rf, &df};
            Act o[2];
            conv2(n, x, 1, 1, ep, dst, o);
        }
        {
            const std::string n[2] = {p + ".r_esa", p + ".d_esa"};
            const Act x[2] = {rd, dr};
            const Act dst[2] = {r_dst, d_dst};
            const Act* add[2] = {residual ? &rgb : nullptr, residual ? &depth : nullptr};
            esa2(n, x, dst, add);
        }
        arena.top = mark;
    }

    // modules/transform/attention.py:63-67: y = x * g (mode 0) or x + x * g (mode 1), g = the per-(n,c) sigmoid weights of x
    // means / mstride: the channel means of x when somebody holds them already (Bi-CEE: SliceMeans), else they are computed.
    void se_scale_to(const std::string& p, const Act& x, int mode, const Act& y, const float* means = nullptr, int mstride = 0)
    {
        float* w0 = dense_of(p + ".fc.0.weight");
        float* w1 = dense_of(p + ".fc.2.weight");
        float* mean = means ? nullptr : (float*)arena.take((size_t)x.n * x.c * sizeof(float));
        float* sc = (float*)arena.take((size_t)x.n * x.c * sizeof(float));
        float* hid = (float*)arena.take((size_t)x.n * (x.c / 16 + 1) * sizeof(float));
        if (dry() || rc || !w0 || !w1) return;
        const int HW = x.h * x.w;
        int r = means ? RGBD_OK : (refnum ? launch_channel_mean_ref(x.p, x.n, HW, x.cs, x.c, mean, x.c, s)
                                          : launch_channel_mean(x.p, x.n, HW, x.cs, x.c, mean, s));
        const float* mu = means ? means : mean;
        if (!r && refnum)
            r = launch_se_fc_ref(mu, x.n, x.c, x.c / 16, w0, w1, cls_of(p + ".fc.0.weight"), cls_of(p + ".fc.2.weight"), hid, sc, s,
                                 means ? mstride : 0, se_form());
        else if (!r) r = launch_se_fc(mu, x.n, x.c, x.c / 16, w0, w1, hid, sc, s, means ? mstride : 0, perm());
        if (!r) r = launch_channel_scale_to(x.p, x.n, HW, x.cs, x.c, sc, mode, y.p, y.cs, s);
        if (r) fail(r);
    }

    // the SE of cat(own, other) written into the two halves of f (synthesis.py:345-362): means side by side, one gate
    void se_cat_to(const std::string& p, const Act& own, const Act& other, const Act& f)
    {
        const int C = own.c + other.c;
        float* w0 = dense_of(p + ".fc.0.weight");
        float* w1 = dense_of(p + ".fc.2.weight");
        float* mean = (float*)arena.take((size_t)own.n * C * sizeof(float));
        float* sc = (float*)arena.take((size_t)own.n * C * sizeof(float));
        float* hid = (float*)arena.take((size_t)own.n * (C / 16 + 1) * sizeof(float));
        if (C % 16 || own.c % 4) {
            fail(RGBD_EINVAL);
            return;
        }
        if (dry() || rc || !w0 || !w1) return;
        const int HW = own.h * own.w;
        int r = refnum ? launch_channel_mean_ref(own.p, own.n, HW, own.cs, own.c, mean, C, s)
                       : launch_channel_mean_strided(own.p, own.n, HW, own.cs, own.c, mean, C, s);
        if (!r)
            r = refnum ? launch_channel_mean_ref(other.p, other.n, HW, other.cs, other.c, mean + own.c, C, s)
                       : launch_channel_mean_strided(other.p, other.n, HW, other.cs, other.c, mean + own.c, C, s);
        if (!r && refnum)
            r = launch_se_fc_ref(mean, own.n, C, C / 16, w0, w1, cls_of(p + ".fc.0.weight"), cls_of(p + ".fc.2.weight"), hid, sc, s, 0, se_form());
        else if (!r) r = launch_se_fc(mean, own.n, C, C / 16, w0, w1, hid, sc, s, 0, perm());
        if (!r) r = launch_channel_scale_to_strided(own.p, own.n, HW, own.cs, own.c, sc, C, 0, f.p, f.cs, s);
        if (!r) r = launch_channel_scale_to_strided(other.p, other.n, HW, other.cs, other.c, sc + own.c, C, 0, f.p + own.c, f.cs, s);
        if (r) fail(r);
    }

    // ---- transforms -----------------------------------------------------------------------------
    // analysis.py:116-174
    void g_a(const Act& rgb_in, const Act& depth_in, Act* y_r, Act* y_d)
    {
        static const char* kinds[18] = {"conv", "rb", "rb", "rb", "spf", "conv", "rb", "rb", "rb",
                                        "attn", "spf", "conv", "rb", "rb", "rb", "spf", "conv", "attn"};
        const std::string pr = "g_a.rgb_analysis_transform.", pd = "g_a.depth_analysis_transform.";
        Act r = rgb_in, d = depth_in;
        Ends ends = ends_begin();
        for (int i = 0; i < 18; ++i) {
            const std::string k = kinds[i], si = std::to_string(i);
            const bool next_spf = (i + 1 < 18) && std::string(kinds[i + 1]) == "spf";
            ends_stage(ends, k != "spf");
            Act rdst, ddst;
            const Act *pr_dst = nullptr, *pd_dst = nullptr;
            Act rcat, dcat;
            if (next_spf) {  // the stage feeding a fusion writes into the first half of the concat buffer
                rcat = alloc(r.n, r.h, r.w, 2 * N);
                dcat = alloc(d.n, d.h, d.w, 2 * N);
                rdst = view(rcat, 0, N);
                ddst = view(dcat, 0, N);
                pr_dst = &rdst;
                pd_dst = &ddst;
            }
            const std::string nm[2] = {pr + si, pd + si};
            const Act xin[2] = {r, d};
            const Act* dsts[2] = {pr_dst, pd_dst};
            Act o[2];
            if (k == "conv") {
                const Epi none[2];
                conv2(nm, xin, 2, 2, none, nullptr, o);
                r = o[0];
                d = o[1];
            } else if (k == "rb") {
                const bool next_rb = (i + 1 < 18) && std::string(kinds[i + 1]) == "rb";
                const std::string sn = std::to_string(i + 1) + ".branch.0";
                const std::string nl[2] = {next_rb ? pr + sn : std::string(), next_rb ? pd + sn : std::string()};
                bottleneck2(nm, xin, dsts, nl, o);
                r = o[0];
                d = o[1];
                if (next_spf) {
                    r = rcat;
                    d = dcat;
                }
            } else if (k == "attn") {
                attention2(nm, xin, dsts, o);
                r = o[0];
                d = o[1];
                if (next_spf) {
                    r = rcat;
                    d = dcat;
                }
            } else {  // spf: r and d are the 2N-channel concat buffers whose first half is filled
                bi_spf(pr + si, view(r, 0, N), view(d, 0, N), view(r, N, N), view(d, N, N));
            }
        }
        ends_finish(ends);
        *y_r = r;
        *y_d = d;
    }

    // synthesis.py:126-184
    void g_s(const Act& yr, const Act& yd, Act* xr, Act* xd)
    {
        static const char* kinds[18] = {"attn", "deconv", "spf", "rb", "rb", "rb", "deconv", "attn", "spf",
                                        "rb", "rb", "rb", "deconv", "spf", "rb", "rb", "rb", "deconv"};
        const std::string pr = "g_s.rgb_synthesis_transform.", pd = "g_s.depth_synthesis_transform.";
        Act r = yr, d = yd;
        Ends ends = ends_begin();
        for (int i = 0; i < 18; ++i) {
            const std::string k = kinds[i], si = std::to_string(i);
            const bool next_spf = (i + 1 < 18) && std::string(kinds[i + 1]) == "spf";
            ends_stage(ends, k != "spf");
            if (k == "spf") {
                bi_spf(pr + si, view(r, 0, N), view(d, 0, N), view(r, N, N), view(d, N, N));
                continue;
            }
            Act rcat, dcat, rdst, ddst;
            const Act *pr_dst = nullptr, *pd_dst = nullptr;
            if (next_spf) {
                const int oh = (k == "deconv") ? r.h * 2 : r.h, ow = (k == "deconv") ? r.w * 2 : r.w;
                rcat = alloc(r.n, oh, ow, 2 * N);
                dcat = alloc(d.n, oh, ow, 2 * N);
                rdst = view(rcat, 0, N);
                ddst = view(dcat, 0, N);
                pr_dst = &rdst;
                pd_dst = &ddst;
            }
            const std::string nm[2] = {pr + si, pd + si};
            const Act xin[2] = {r, d};
            const Act* dsts[2] = {pr_dst, pd_dst};
            Act o[2];
            if (k == "deconv") {
                const Epi none[2];
                conv2(nm, xin, 2, 2, none, next_spf ? dsts : nullptr, o);
            } else if (k == "rb") {
                const bool next_rb = (i + 1 < 18) && std::string(kinds[i + 1]) == "rb";
                const std::string sn = std::to_string(i + 1) + ".branch.0";
                const std::string nl[2] = {next_rb ? pr + sn : std::string(), next_rb ? pd + sn : std::string()};
                bottleneck2(nm, xin, dsts, nl, o);
            } else {
                attention2(nm, xin, dsts, o);
            }
            r = o[0];
            d = o[1];
            if (next_spf) {
                r = rcat;
                d = dcat;
            }
        }
        ends_finish(ends);
        *xr = r;
        *xd = d;
    }

    // analysis.py:231-242
    void h_a(const Act& yr, const Act& yd, Act* zr, Act* zd)
    {
        Epi relu;
        relu.act = ACT_RELU;
        const char* mods[2] = {"rgb", "depth"};
        const Act* in[2] = {&yr, &yd};
        Act* out[2] = {zr, zd};
        const std::string p[2] = {std::string("h_a.") + mods[0] + "_reduction.", std::string("h_a.") + mods[1] + "_reduction."};
        const Epi relu2[2] = {relu, relu}, none[2];
        const Act x0[2] = {*in[0], *in[1]};
        Act t0[2], t1[2], t2[2];
        const std::string n0[2] = {p[0] + "0", p[1] + "0"}, n2[2] = {p[0] + "2", p[1] + "2"}, n4[2] = {p[0] + "4", p[1] + "4"};
        conv2(n0, x0, 1, 1, relu2, nullptr, t0);
        conv2(n2, t0, 2, 2, relu2, nullptr, t1);
        conv2(n4, t1, 2, 2, none, nullptr, t2);
        *out[0] = t2[0];
        *out[1] = t2[1];
    }

    // conv_transpose2d(k 5, stride 2, pad 2, output_padding 1) + activation in the reference's CPU arithmetic (oneDNN's
    // brg_deconv: DESIGN.md 4a; oracle/cpu_arith.c orc_deconv_s2): the taps of an output pixel are accumulated tap by tap over
    // all input channels, in chains whose membership depends on the layer shape and on the pixel's column block -- a measured
    // recipe per (phase, column), refarith_tables.json kind 3.  Per phase and column class (columns with the same recipe) this
    // is ONE GEMM whose K axis is (tap, channel): gathered input rows x gathered weight slabs, the chains as split-K ranges
    // whose sums the ordered reducer adds (+ bias, activation), rows scattered to the phase's output positions.
    // false = no recipe for this shape (the caller runs the sub-pixel-phase kernel).
    bool deconv_s2_ref(const std::string& name, const Act& x, int act, Act* out)
    {
        if (!refnum) return false;
        const PackedConv* pc = conv_of(name + ".weight");
        if (!pc || !pc->transposed || pc->k != 5 || pc->subpix || x.cs != pc->cin_pad) return false;
        const std::vector<int>* rec = ref_blocks(3, pc->cin, pc->cout, x.h, x.w);
        if (!rec) return false;
        const int h = x.h, w = x.w, B = x.n;
        *out = alloc(B, 2 * h, 2 * w, pc->cout);
        // parse: 4 * w descriptors {n, n x (ky, kx, fresh)}
        std::vector<const int*> desc(4 * (size_t)w, nullptr);
        {
            size_t pos = 0;
            for (size_t i = 0; i < desc.size(); ++i) {
                if (pos >= rec->size()) {
                    fail(RGBD_EINVAL);
                    return true;
                }
                desc[i] = rec->data() + pos;
                pos += 1 + 3 * (size_t)(*rec)[pos];
            }
            if (pos != rec->size()) {
                fail(RGBD_EINVAL);
                return true;
            }
        }
        auto same = [](const int* a, const int* b) { return a[0] == b[0] && memcmp(a, b, sizeof(int) * (1 + 3 * (size_t)a[0])) == 0; };
        for (int ph = 0; ph < 4; ++ph) {
            const int py = ph >> 1, px = ph & 1;
            for (int j0 = 0; j0 < w;) {
                const int* d = desc[(size_t)ph * w + j0];
                int j1 = j0 + 1;
                while (j1 < w && same(d, desc[(size_t)ph * w + j1])) ++j1;
                const int jw = j1 - j0, nt = d[0];
                if (nt < 1 || nt > 16) {
                    fail(RGBD_EINVAL);
                    return true;
                }
                int dy[16], dx[16], slab[16], nch = 0;
                uint16_t bnd[18] = {0};
                for (int t = 0; t < nt; ++t) {
                    const int ky = d[1 + 3 * t], kx = d[2 + 3 * t], fresh = d[3 + 3 * t];
                    dy[t] = (py + 2 - ky) / 2;  // input row of output row 2 ty + py under tap ky: ty + (py + pad - ky) / 2
                    dx[t] = (px + 2 - kx) / 2;
                    slab[t] = ky * 5 + kx;
                    if (fresh || t == 0) bnd[nch++] = (uint16_t)(t * (pc->cin_pad / 16));
                }
                bnd[nch] = (uint16_t)(nt * (pc->cin_pad / 16));
                if (nch > 16) {
                    fail(RGBD_EINVAL);
                    return true;
                }
                const size_t mark = arena.top;
                const size_t npx = (size_t)B * h * jw;
                const int Kp = nt * pc->cin_pad;
                float* col = (float*)arena.take(npx * Kp * sizeof(float));
                float* wsel = (float*)arena.take((size_t)pc->cout_pad * Kp * sizeof(float));
                float* tmp = (float*)arena.take(npx * pc->cout_pad * sizeof(float));
                float* part = nch > 1 ? (float*)arena.take((size_t)nch * npx * pc->cout_pad * sizeof(float)) : nullptr;
                if (!dry() && !rc) {
                    int r = launch_gather_taps(x.p, B, h, w, x.cs, j0, jw, nt, dy, dx, col, s);
                    if (!r) r = launch_gather_wslabs(pc->w, pc->cout_pad, 25, pc->cin_pad, nt, slab, wsel, s);
                    if (!r) {
                        ConvArgs a{};
                        a.x = col;
                        a.N = 1;
                        a.H = B * h;
                        a.W = jw;
                        a.xcs = Kp;
                        a.cin_pad = Kp;
                        a.w = wsel;
                        a.ntaps_total = 1;
                        a.bias = pc->bias;
                        a.y = tmp;
                        a.OH = B * h;
                        a.OW = jw;
                        a.ycs = pc->cout_pad;
                        a.cout_pad = pc->cout_pad;
                        a.cout_store = pc->cout_pad;
                        a.GH = B * h;
                        a.GW = jw;
                        a.IS = a.OS = 1;
                        a.nphase = 1;
                        a.taps.n[0] = 1;
                        a.span_y = a.span_x = 1;
                        a.act = act;
                        a.loaded = tile_mode;
                        a.exact_math = 1;
                        a.splitk = nch;
                        if (nch > 1) {
                            a.partial = part;
                            for (int c = 0; c <= nch; ++c) a.split_c16[c] = bnd[c];
                        }
                        r = launch_conv(a, s);
                    }
                    if (!r) r = launch_scatter_phase(tmp, B, h, jw, pc->cout_pad, j0, py, px, out->p, 2 * w, out->cs, pc->cout_pad, s);
                    if (r) fail(r);
                }
                arena.top = mark;
                j0 = j1;
            }
        }
        return true;
    }

    // synthesis.py:345-362.  cat(own, other) -> SE -> deconv without materialising the unscaled concatenation: the channel
    // means of the two inputs land side by side (what the mean of the concatenation would be, channel by channel), the
    // gate is computed from them, and each input is scaled straight into its half of the deconv's input buffer
    Act hs_block(const std::string& p, const Act& own, const Act& other, bool last)
    {
        Act f = alloc(own.n, own.h, own.w, own.c + other.c);
        se_cat_to(p + ".se", own, other, f);
        Epi e;
        e.act = last ? ACT_NONE : ACT_LEAKY;
        Act o;
        if (!last && deconv_s2_ref(p + ".deconv", f, ACT_LEAKY, &o)) return o;
        return conv(p + ".deconv", f, last ? 1 : 2, last ? 1 : 2, e);
    }

    // one stage of both modalities: the SE gates stay per modality, the two (transposed) convs are one grouped launch
    void hs_block2(const std::string p[2], const Act own[2], const Act other[2], bool last, Act out[2])
    {
        Act f[2];
        for (int m = 0; m < 2; ++m) {
            f[m] = alloc(own[m].n, own[m].h, own[m].w, own[m].c + other[m].c);
            se_cat_to(p[m] + ".se", own[m], other[m], f[m]);
            if (rc) return;
        }
        Epi e[2];
        e[0].act = e[1].act = last ? ACT_NONE : ACT_LEAKY;
        const std::string n[2] = {p[0] + ".deconv", p[1] + ".deconv"};
        const PackedConv* pc0 = conv_of(n[0] + ".weight");
        if (!last && refnum && pc0 && ref_blocks(3, pc0->cin, pc0->cout, f[0].h, f[0].w)) {
            bool ok = true;
            for (int m = 0; m < 2; ++m) ok = deconv_s2_ref(n[m], f[m], ACT_LEAKY, &out[m]) && ok;
            if (!ok) fail(RGBD_ESTATE);
            return;
        }
        conv2(n, f, last ? 1 : 2, last ? 1 : 2, e, nullptr, out);
    }

    // synthesis.py:316-323
    void h_s(const Act& zr, const Act& zd, Act* hr, Act* hd)
    {
        Act cur[2] = {zr, zd};
        for (int st = 1; st <= 3; ++st) {
            const std::string p[2] = {"h_s.r_h_s" + std::to_string(st), "h_s.d_h_s" + std::to_string(st)};
            const Act other[2] = {cur[1], cur[0]};
            Act o[2];
            hs_block2(p, cur, other, st == 3, o);
            cur[0] = o[0];
            cur[1] = o[1];
        }
        *hr = cur[0];
        *hd = cur[1];
    }

    // entropy.py:69-78.  `ctx` is a channel-slice view of the slice's context buffer
    // [r_loc | d_loc | hyper_r | hyper_d | ch_ctx_r | ch_ctx_d]: every EntropyParametersEX input of the reference
    // (elic_united.py:288-333) is a suffix of that layout, so no concatenation copy is needed; SE-rescaling writes the
    // rescaled copy the 1x1 conv reads (params + se(params), keeping the reference's association).
    // `part` (1 anchor / 2 non-anchor): the caller only reads that checkerboard half of (scales, means)
    // (ckbd.py:83-125), so the last -- and largest -- conv computes just that half; the values are those of the full conv.
    Act entropy_params(const std::string& p, const Act& ctx, int part, const Act* dst = nullptr, const float* means = nullptr,
                       int mstride = 0)
    {
        const PackedConv* last = conv_of(p + ".fusion.4.weight");
        if (!last) return Act();
        Act out = dst ? *dst : alloc(ctx.n, ctx.h, ctx.w, last->cout);
        const size_t mark = arena.top;
        Act cat = alloc(ctx.n, ctx.h, ctx.w, ctx.c);
        se_scale_to(p + ".se", ctx, 1, cat, means, mstride);
        Epi relu;
        relu.act = ACT_RELU;
        Act t = conv(p + ".fusion.0", cat, 1, 0, relu);
        t = conv(p + ".fusion.2", t, 1, 1, relu);
        Epi last_e;
        last_e.ckbd = g_ckbd_conv ? part : 0;
        conv(p + ".fusion.4", t, 1, 2, last_e, &out);
        arena.top = mark;
        return out;
    }

    // context.py:10-30
    Act channel_context(const std::string& p, const Act& x, const Act* dst = nullptr)
    {
        const PackedConv* last = conv_of(p + ".fushion.4.weight");
        if (!last) return Act();
        Act out = dst ? *dst : alloc(x.n, x.h, x.w, last->cout);
        const size_t mark = arena.top;
        Epi relu;
        relu.act = ACT_RELU;
        Act t = conv(p + ".fushion.0", x, 1, 2, relu);
        t = conv(p + ".fushion.2", t, 1, 2, relu);
        conv(p + ".fushion.4", t, 1, 2, Epi(), &out);
        arena.top = mark;
        return out;
    }

    // context.py:10-30 for both modalities (slice i's two nets read only what earlier slices decoded: independent)
    void channel_context2(const std::string p[2], const Act x[2], const Act dst[2])
    {
        const size_t mark = arena.top;
        Epi relu[2];
        relu[0].act = relu[1].act = ACT_RELU;
        const Epi none[2];
        const std::string n0[2] = {p[0] + ".fushion.0", p[1] + ".fushion.0"}, n2[2] = {p[0] + ".fushion.2", p[1] + ".fushion.2"},
                          n4[2] = {p[0] + ".fushion.4", p[1] + ".fushion.4"};
        Act t0[2], t1[2], o[2];
        conv2(n0, x, 1, 2, relu, nullptr, t0);
        conv2(n2, t0, 1, 2, relu, nullptr, t1);
        const Act* odst[2] = {&dst[0], &dst[1]};
        conv2(n4, t1, 1, 2, none, odst, o);
        arena.top = mark;
    }

    // ---- Bi-CEE loop (elic_united.py:265-348 / 454-541) -----------------------------------------
    struct Coding {
        bool encode = true;
        bool estimate = false;      // eval-mode forward(): quantise + likelihood, no symbols
        Act lik[2];                 // likelihood tensors [B,h,w,M] per modality (estimate mode)
        int per_image = 1;
        int64_t per_image_total = 0;  // symbols per image per modality
        int32_t* sym = nullptr;       // [2][B*per_image_total]
        int32_t* idx = nullptr;
        const int64_t* stream_base = nullptr;  // device [B] symbol base of each stream inside a modality region
        const int32_t* force = nullptr;        // teacher forcing: [2][B*per_image_total] symbols later contexts are built from
        // decode side
        const uint32_t* words = nullptr;
        const int64_t* stream_off = nullptr;  // device [2][nstreams]
        const int64_t* stream_len = nullptr;
        uint64_t* state = nullptr;  // device [2][nstreams][2]
        int nstreams = 0;
        bool first[2] = {true, true};
    };

    void code_part(Coding& cd, int mod, int anchor, const Act& params, const Act& y_slice, const Act& yhat_slice,
                   int64_t part_off)
    {
        if (dry() || rc) return;
        PartGeom g;
        g.B = params.n;
        g.h = params.h;
        g.w = params.w;
        g.C = yhat_slice.c;
        g.anchor = anchor;
        g.per_image = cd.per_image;
        g.perm = perm();
        const int64_t mod_off = (int64_t)mod * g.B * cd.per_image_total;
        int32_t* sym = cd.sym + mod_off;
        int32_t* idx = cd.idx + mod_off;
        const int64_t* sb = cd.stream_base;  // relative to the modality's region
        int r;
        if (cd.estimate) {
            const Act lk = view(cd.lik[mod], (int)(yhat_slice.p - (mod ? yhat_base[1] : yhat_base[0])), g.C);
            r = launch_ckbd_estimate_part(y_slice.p, y_slice.cs, params.p, params.cs, yhat_slice.p, yhat_slice.cs, lk.p, lk.cs,
                                          g, s);
        } else if (cd.encode) {
            r = launch_ckbd_encode_part(y_slice.p, y_slice.cs, params.p, params.cs, yhat_slice.p, yhat_slice.cs,
                                        scale_table, g, sym, idx, sb, part_off, s, dbg_x ? dbg_x + mod_off : nullptr,
                                        dbg_s ? dbg_s + mod_off : nullptr);
            if (!r && cd.force)  // y_hat of this part again, from the forced symbols (symbol + mean, as the decoder forms it)
                r = launch_ckbd_decode_part(params.p, params.cs, yhat_slice.p, yhat_slice.cs, g, cd.force + mod_off, sb, part_off, s);
        } else {
            r = launch_ckbd_index_part(params.p, params.cs, scale_table, g, idx, sb, part_off, s);
            const int64_t count = (int64_t)g.C * g.h * (g.w / 2) * (cd.per_image ? 1 : g.B);
            const int64_t poff = cd.per_image ? part_off : part_off * g.B;
            if (!r)
                r = launch_rans_decode(cd.words, cd.stream_off + (size_t)mod * cd.nstreams,
                                       cd.stream_len + (size_t)mod * cd.nstreams, cd.nstreams,
                                       cd.state + (size_t)mod * cd.nstreams * 2, cd.first[mod] ? 1 : 0, idx, sym, sb, poff,
                                       count, tables[mod].d, s);
            cd.first[mod] = false;
            if (!r) r = launch_ckbd_decode_part(params.p, params.cs, yhat_slice.p, yhat_slice.cs, g, sym, sb, part_off, s);
        }
        if (r) fail(r);
    }

    float* yhat_base[2] = {nullptr, nullptr};

    void bicee(Coding& cd, const Act* y_r, const Act* y_d, const Act& hyp_r, const Act& hyp_d, const Act& yhat_r,
               const Act& yhat_d)
    {
        yhat_base[0] = yhat_r.p;
        yhat_base[1] = yhat_d.p;
        int c0 = 0;
        int64_t part_off = 0;
        const int B = hyp_r.n, h = hyp_r.h, w = hyp_r.w;
        // SE gates of the entropy-parameter nets (entropy.py:75) need the channel means of their whole input -- 1280 ... 2816
        // channels, of which 2 x 2M are the hyper parameters, the same tensor for all 20 nets of a call.  A mean is a function
        // of its own channel only (channel_mean_kernel: one fixed chain per channel), so the means are kept per segment of
        // the context buffer and only what changed is recomputed: the hyper parameters' once per call, the channel contexts'
        // once per slice, the local contexts' (2C channels) per part -- the same floats as a pass over the whole input, for
        // 1/10 of the traffic (round 4; 1.6 GB per c3 step).  hm: [B][2 HC] hyper means; sm: [B][wide] in ctx layout.
        static const bool mean_cache = getenv("RGBD_NO_MEAN_CACHE") == nullptr;  // A/B switch
        const int HC2 = 2 * hyp_r.c;
        float* hm = (float*)arena.take((size_t)B * HC2 * sizeof(float));
        auto means_of = [&](const Act& t, float* dstm, int stride) {
            if (dry() || rc || !mean_cache) return;
            const int r = refnum ? launch_channel_mean_ref(t.p, t.n, t.h * t.w, t.cs, t.c, dstm, stride, s)
                                 : launch_channel_mean_strided(t.p, t.n, t.h * t.w, t.cs, t.c, dstm, stride, s);
            if (r) fail(r);
        };
        means_of(hyp_r, hm, HC2);
        means_of(hyp_d, hm + hyp_r.c, HC2);
        for (size_t i = 0; i < slice_ch.size(); ++i) {
            const int C = slice_ch[i];
            const size_t mark = arena.top;
            const std::string si = std::to_string(i);
            // context buffer of this slice: [r_loc 2C | d_loc 2C | hyper_r 2M | hyper_d 2M | ch_r 2C | ch_d 2C]
            const int HC = hyp_r.c;  // 2M
            const int wide = 4 * C + 2 * HC + (i ? 4 * C : 0);
            Act ctx = alloc(hyp_r.n, h, w, wide);
            float* sm = (float*)arena.take((size_t)B * wide * sizeof(float));
            copy_ch(hyp_r, view(ctx, 4 * C, HC));
            copy_ch(hyp_d, view(ctx, 4 * C + HC, HC));
            if (!dry() && !rc && mean_cache) {
                const int r = launch_copy_channels(hm, HC2, sm + 4 * C, wide, B, HC2, s);
                if (r) fail(r);
            }
            if (i) {
                const Act cr = view(ctx, 4 * C + 2 * HC, 2 * C), cdv = view(ctx, 6 * C + 2 * HC, 2 * C);
                const std::string cn[2] = {"rgb_channel_context." + si, "depth_channel_context." + si};
                const Act cx[2] = {view(yhat_r, 0, c0), view(yhat_d, 0, c0)};
                const Act cdst[2] = {cr, cdv};
                channel_context2(cn, cx, cdst);
                means_of(view(ctx, 4 * C + 2 * HC, 4 * C), sm + 4 * C + 2 * HC, wide);  // both channel contexts: adjacent
            }
            const float* smc = mean_cache ? sm : nullptr;
            const Act yr = y_r ? view(*y_r, c0, C) : Act();
            const Act yd = y_d ? view(*y_d, c0, C) : Act();
            const Act hr = view(yhat_r, c0, C), hd = view(yhat_d, c0, C);
            const int64_t part_syms = (int64_t)C * h * (w / 2);
            const Act r_loc = view(ctx, 0, 2 * C), d_loc = view(ctx, 2 * C, 2 * C);
            // rgb anchor: [hyper, ch ctx]
            Act p_ra = entropy_params("rgb_entropy_parameters_anchor." + si, view(ctx, 4 * C, wide - 4 * C), 1, nullptr,
                                      smc ? smc + 4 * C : nullptr, wide);
            code_part(cd, 0, 1, p_ra, yr, hr, part_off);
            conv_anchor_in("rgb_local_context." + si, hr, 2, r_loc);  // (hr holds the anchor half only so far)
            means_of(r_loc, sm, wide);
            // depth anchor: [r_loc, hyper, ch ctx] -- d_loc's slot sits between them, so this one input is gathered
            Act p_da = alloc(hyp_r.n, h, w, 2 * C);
            {
                const size_t m2 = arena.top;
                Act in = alloc(hyp_r.n, h, w, wide - 2 * C);
                copy_ch(r_loc, view(in, 0, 2 * C));
                copy_ch(view(ctx, 4 * C, wide - 4 * C), view(in, 2 * C, wide - 4 * C));
                float* im = (float*)arena.take((size_t)B * (wide - 2 * C) * sizeof(float));  // the gathered input's means, gathered alike
                if (!dry() && !rc && mean_cache) {
                    int r = launch_copy_channels(sm, wide, im, wide - 2 * C, B, 2 * C, s);
                    if (!r) r = launch_copy_channels(sm + 4 * C, wide, im + 2 * C, wide - 2 * C, B, wide - 4 * C, s);
                    if (r) fail(r);
                }
                entropy_params("depth_entropy_parameters_anchor." + si, in, 1, &p_da, mean_cache ? im : nullptr, wide - 2 * C);
                arena.top = m2;
            }
            code_part(cd, 1, 1, p_da, yd, hd, part_off);
            conv_anchor_in("depth_local_context." + si, hd, 2, d_loc);
            means_of(d_loc, sm + 2 * C, wide);
            // rgb non-anchor: the whole buffer
            Act p_rn = entropy_params("rgb_entropy_parameters_nonanchor." + si, ctx, 2, nullptr, smc, wide);
            code_part(cd, 0, 0, p_rn, yr, hr, part_off + part_syms);
            conv("rgb_local_context_anchor_with_nonanchor." + si, hr, 1, 2, Epi(), &r_loc);  // replaces r_loc
            means_of(r_loc, sm, wide);
            // depth non-anchor
            Act p_dn = entropy_params("depth_entropy_parameters_nonanchor." + si, ctx, 2, nullptr, smc, wide);
            code_part(cd, 1, 0, p_dn, yd, hd, part_off + part_syms);
            part_off += 2 * part_syms;
            c0 += C;
            arena.top = mark;
        }
    }

    // ---- STF_united (models/stf_united.py; BASELINE config 5): Swin transforms on [B,H,W,C] token maps -------------
    Act layernorm(const std::string& p, const Act& x)
    {
        Act y = alloc(x.n, x.h, x.w, x.c);
        float* w = dense_of(p + ".weight");
        float* b = dense_of(p + ".bias");
        if (dry() || rc || !w || !b) return y;
        const int r = launch_layernorm(x.p, (size_t)x.n * x.h * x.w, x.c, x.cs, w, b, y.p, y.cs, s);
        if (r) fail(r);
        return y;
    }
    // stf_united.py:118-214: x + proj(attn(norm1(x))), then + mlp(norm2(.)); GELU and both adds are conv epilogues -- for both
    // modalities at once (round 4): the RGB and the depth stack of STF_united run the same layer shapes
    // on independent data between two fusions, so every Linear is one grouped conv launch (conv2) and every LayerNorm /
    // window attention one launch over both tensors -- half the launches of a model whose launches are too small to fill the
    // chip (35 us on average at one 512x512 pair).  Each output keeps its arithmetic: bit-identical to the one-by-one form.
    void layernorm2(const std::string p[2], const Act x[2], Act y[2])
    {
        float *w[2], *b[2];
        for (int m = 0; m < 2; ++m) {
            y[m] = alloc(x[m].n, x[m].h, x[m].w, x[m].c);
            w[m] = dense_of(p[m] + ".weight");
            b[m] = dense_of(p[m] + ".bias");
        }
        if (dry() || rc || !w[0] || !b[0] || !w[1] || !b[1]) return;
        const bool same = g_pair && x[0].n == x[1].n && x[0].h == x[1].h && x[0].w == x[1].w && x[0].c == x[1].c &&
                          x[0].cs == x[1].cs && y[0].cs == y[1].cs;
        const size_t ntok = (size_t)x[0].n * x[0].h * x[0].w;
        int r = launch_layernorm(x[0].p, ntok, x[0].c, x[0].cs, w[0], b[0], y[0].p, y[0].cs, s, same ? x[1].p : nullptr,
                                 same ? w[1] : nullptr, same ? b[1] : nullptr, same ? y[1].p : nullptr);
        if (!r && !same)
            r = launch_layernorm(x[1].p, (size_t)x[1].n * x[1].h * x[1].w, x[1].c, x[1].cs, w[1], b[1], y[1].p, y[1].cs, s);
        if (r) fail(r);
    }
    void swin_block2(const std::string p[2], const Act x[2], int shift, int heads, Act out[2])
    {
        for (int m = 0; m < 2; ++m) out[m] = alloc(x[m].n, x[m].h, x[m].w, x[m].c);
        const size_t mark = arena.top;
        auto names = [&](const char* suf, std::string n[2]) {
            n[0] = p[0] + suf;
            n[1] = p[1] + suf;
        };
        std::string n[2];
        const Epi none[2];
        Act t[2], qkv[2], a[2], x1[2], t2[2], hdn[2], o[2];
        names(".norm1", n);
        layernorm2(n, x, t);
        names(".attn.qkv", n);
        conv2(n, t, 1, 0, none, nullptr, qkv);
        float* rpb[2];
        for (int m = 0; m < 2; ++m) {
            a[m] = alloc(x[m].n, x[m].h, x[m].w, x[m].c);
            rpb[m] = dense_of(p[m] + ".attn.relative_position_bias_table");
        }
        if (!dry() && !rc && rpb[0] && rpb[1]) {
            const bool same = g_pair && x[0].n == x[1].n && x[0].h == x[1].h && x[0].w == x[1].w && x[0].c == x[1].c &&
                              qkv[0].cs == qkv[1].cs && a[0].cs == a[1].cs;
            int r = launch_window_attention(qkv[0].p, x[0].n, x[0].h, x[0].w, x[0].c, qkv[0].cs, heads, shift, rpb[0], a[0].p, a[0].cs,
                                            s, same ? qkv[1].p : nullptr, same ? rpb[1] : nullptr, same ? a[1].p : nullptr);
            if (!r && !same)
                r = launch_window_attention(qkv[1].p, x[1].n, x[1].h, x[1].w, x[1].c, qkv[1].cs, heads, shift, rpb[1], a[1].p,
                                            a[1].cs, s);
            if (r) fail(r);
        }
        Epi e1[2];
        e1[0].res1 = &x[0];
        e1[1].res1 = &x[1];
        names(".attn.proj", n);
        conv2(n, a, 1, 0, e1, nullptr, x1);
        names(".norm2", n);
        layernorm2(n, x1, t2);
        Epi g[2];
        g[0].act = g[1].act = ACT_GELU;
        names(".mlp.fc1", n);
        conv2(n, t2, 1, 0, g, nullptr, hdn);
        Epi e2[2];
        e2[0].res1 = &x1[0];
        e2[1].res1 = &x1[1];
        const Act* odst[2] = {&out[0], &out[1]};
        names(".mlp.fc2", n);
        conv2(n, hdn, 1, 0, e2, odst, o);
        arena.top = mark;
    }
    // stf_united.py:270-366 for both modalities; down: 0 none, 1 PatchMerging (:217-249), 2 PatchSplit (:252-267)
    void basic_layer2(const std::string p[2], const Act x_in[2], int depth, int heads, int down, Act out[2])
    {
        Act x[2] = {x_in[0], x_in[1]};
        for (int k = 0; k < depth; ++k) {
            const std::string pb[2] = {p[0] + ".blocks." + std::to_string(k), p[1] + ".blocks." + std::to_string(k)};
            Act o[2];
            swin_block2(pb, x, (k & 1) ? 2 : 0, heads, o);
            x[0] = o[0];
            x[1] = o[1];
        }
        const std::string pn[2] = {p[0] + ".downsample.norm", p[1] + ".downsample.norm"};
        const std::string prd[2] = {p[0] + ".downsample.reduction", p[1] + ".downsample.reduction"};
        const Epi none[2];
        if (down == 1) {
            Act g4[2], t[2];
            for (int m = 0; m < 2; ++m) {
                g4[m] = alloc(x[m].n, x[m].h / 2, x[m].w / 2, 4 * x[m].c);
                if (!dry() && !rc) {
                    const int r = launch_patch_merge_gather(x[m].p, x[m].n, x[m].h, x[m].w, x[m].c, x[m].cs, g4[m].p, g4[m].cs, s);
                    if (r) fail(r);
                }
            }
            layernorm2(pn, g4, t);
            conv2(prd, t, 1, 0, none, nullptr, out);
            return;
        }
        if (down == 2) {
            Act t[2], r2[2];
            layernorm2(pn, x, t);
            conv2(prd, t, 1, 0, none, nullptr, r2);
            for (int m = 0; m < 2; ++m) {
                out[m] = alloc(x[m].n, 2 * x[m].h, 2 * x[m].w, x[m].c / 2);
                if (!dry() && !rc) {
                    const int r = launch_pixel_shuffle2(r2[m].p, x[m].n, x[m].h, x[m].w, x[m].c / 2, r2[m].cs, out[m].p, out[m].cs, s);
                    if (r) fail(r);
                }
            }
            return;
        }
        out[0] = x[0];
        out[1] = x[1];
    }
    void stf_stack(const std::string& root, const char* kind, const Act& r_in, const Act& d_in, const int* depths,
                   const int* heads, int down, Act* r_out, Act* d_out)
    {
        Act r = r_in, d = d_in;
        int li = 0;
        for (int i = 0; i < 4; ++i) {
            const int dn = i < 3 ? down : 0;
            const std::string pl[2] = {root + ".rgb_" + kind + "_layers." + std::to_string(li),
                                       root + ".depth_" + kind + "_layers." + std::to_string(li)};
            const Act xin[2] = {r, d};
            Act o[2];
            basic_layer2(pl, xin, depths[i], heads[i], dn, o);
            r = o[0];
            d = o[1];
            ++li;
            if (i < 3) {  // Bi-CPT fusion added to the streams (stf_united.py:481-489 / 581-589)
                Act r2 = alloc(r.n, r.h, r.w, r.c), d2 = alloc(d.n, d.h, d.w, d.c);
                bi_spf(root + ".rgb_" + kind + "_layers." + std::to_string(li), r, d, r2, d2, true);
                r = r2;
                d = d2;
                ++li;
            }
        }
        *r_out = r;
        *d_out = d;
    }
    void g_a_stf(const Act& rgb, const Act& depth, Act* y_r, Act* y_d)
    {
        static const int depths[4] = {2, 2, 6, 2}, heads[4] = {3, 6, 12, 24};
        Act r = layernorm("g_a.rgb_patch_embed.norm", conv("g_a.rgb_patch_embed.proj", rgb, 2, 0));
        Act d = layernorm("g_a.depth_patch_embed.norm", conv("g_a.depth_patch_embed.proj", depth, 2, 0));
        stf_stack("g_a", "ana", r, d, depths, heads, 1, y_r, y_d);
    }
    void g_s_stf(const Act& yr, const Act& yd, Act* xr, Act* xd)
    {
        static const int depths[4] = {2, 6, 2, 2}, heads[4] = {24, 12, 6, 3};
        Act r, d;
        stf_stack("g_s", "syn", yr, yd, depths, heads, 2, &r, &d);
        // stf_united.py:550-559; the first end conv has the same shape in both modalities (one grouped launch), the last differs
        const std::string n0[2] = {"g_s.rgb_end_conv.0", "g_s.depth_end_conv.0"};
        const Act in[2] = {r, d};
        const Epi none[2];
        Act t[2];
        conv2(n0, in, 1, 2, none, nullptr, t);
        const char* mods[2] = {"rgb", "depth"};
        Act* out[2] = {xr, xd};
        for (int m = 0; m < 2; ++m) {
            Act u = alloc(t[m].n, 2 * t[m].h, 2 * t[m].w, t[m].c / 4);
            if (!dry() && !rc) {
                const int q = launch_pixel_shuffle2(t[m].p, t[m].n, t[m].h, t[m].w, t[m].c / 4, t[m].cs, u.p, u.cs, s);
                if (q) fail(q);
            }
            *out[m] = conv(std::string("g_s.") + mods[m] + "_end_conv.2", u, 1, 1);
        }
    }

    // ---- ELIC_united_R2D (models/elic_united_R2D.py): RGB on its own, depth conditioned on RGB ------
    // attention.py:14-32: only the depth-side gated features exist
    void bi_spf_single(const std::string& p, const Act& rgb, const Act& depth, const Act& d_dst)
    {
        const size_t mark = arena.top;
        const int half = rgb.c / 2;
        Act dr = alloc(rgb.n, rgb.h, rgb.w, rgb.c);  // cat(df, rf)
        Epi relu;
        relu.act = ACT_RELU;
        Act df = view(dr, 0, half), rf = view(dr, half, half);
        conv(p + ".d_ext", depth, 1, 1, relu, &df);
        conv(p + ".r_ext", rgb, 1, 1, relu, &rf);
        esa(p + ".d_esa", dr, d_dst);
        arena.top = mark;
    }
    // analysis.py:56-112 / synthesis.py:186-242: the same 18 stages as ELIC_united; the fusion stage only widens depth
    void stack_r2d(const std::string& root, const char* kind, const char* const* kinds, const Act& r_in, const Act& d_in,
                   Act* r_out, Act* d_out)
    {
        const std::string pr = root + ".rgb_" + kind + "_transform.", pd = root + ".depth_" + kind + "_transform.";
        Act r = r_in, d = d_in;
        for (int i = 0; i < 18; ++i) {
            const std::string k = kinds[i], si = std::to_string(i);
            const bool next_spf = (i + 1 < 18) && std::string(kinds[i + 1]) == "spf";
            if (k == "spf") {  // d is the 2N-channel concat buffer whose first half is filled
                bi_spf_single(pr + si, r, view(d, 0, N), view(d, N, N));
                continue;
            }
            Act dcat, ddst;
            const Act* pd_dst = nullptr;
            if (next_spf) {
                const int oh = (k == "deconv") ? d.h * 2 : (k == "conv" ? d.h / 2 : d.h);
                const int ow = (k == "deconv") ? d.w * 2 : (k == "conv" ? d.w / 2 : d.w);
                dcat = alloc(d.n, oh, ow, 2 * N);
                ddst = view(dcat, 0, N);
                pd_dst = &ddst;
            }
            if (k == "conv" || k == "deconv") {
                r = conv(pr + si, r, 2, 2);
                d = conv(pd + si, d, 2, 2, Epi(), pd_dst);
            } else if (k == "rb") {
                r = bottleneck(pr + si, r);
                d = bottleneck(pd + si, d, pd_dst);
            } else {
                r = attention(pr + si, r);
                d = attention(pd + si, d, pd_dst);
            }
            if (next_spf) d = dcat;
        }
        *r_out = r;
        *d_out = d;
    }
    void g_a_r2d(const Act& rgb, const Act& depth, Act* y_r, Act* y_d)
    {
        static const char* const kinds[18] = {"conv", "rb", "rb", "rb", "spf", "conv", "rb", "rb", "rb",
                                              "attn", "spf", "conv", "rb", "rb", "rb", "spf", "conv", "attn"};
        stack_r2d("g_a", "analysis", kinds, rgb, depth, y_r, y_d);
    }
    void g_s_r2d(const Act& yr, const Act& yd, Act* xr, Act* xd)
    {
        static const char* const kinds[18] = {"attn", "deconv", "spf", "rb", "rb", "rb", "deconv", "attn", "spf",
                                              "rb", "rb", "rb", "deconv", "spf", "rb", "rb", "rb", "deconv"};
        stack_r2d("g_s", "synthesis", kinds, yr, yd, xr, xd);
    }
    // synthesis.py:364-380
    Act hs_block_single(const std::string& p, const Act& x, bool last)
    {
        Act f = alloc(x.n, x.h, x.w, x.c);
        se_scale_to(p + ".se", x, 0, f);
        Epi e;
        e.act = last ? ACT_NONE : ACT_LEAKY;
        Act o;
        if (!last && deconv_s2_ref(p + ".deconv", f, ACT_LEAKY, &o)) return o;
        return conv(p + ".deconv", f, last ? 1 : 2, last ? 1 : 2, e);
    }
    // synthesis.py:336-343
    void h_s_r2d(const Act& zr, const Act& zd, Act* hr, Act* hd)
    {
        Act r1 = hs_block_single("h_s.r_h_s1", zr, false);
        Act d1 = hs_block("h_s.d_h_s1", zd, zr, false);
        Act r2 = hs_block_single("h_s.r_h_s2", r1, false);
        Act d2 = hs_block("h_s.d_h_s2", d1, r1, false);
        *hr = hs_block_single("h_s.r_h_s3", r2, true);
        *hd = hs_block("h_s.d_h_s3", d2, r2, true);
    }
    // elic_united_R2D.py:149-326.  RGB context buffer [r_loc 2C | hyper_r 2M | ch_r 2C]: anchor reads the suffix, non-anchor
    // the whole.  Depth context buffer as in ELIC_united: [r_loc 2C | d_loc 2C | hyper_r | hyper_d | ch_r | ch_d].
    void bicee_r2d(Coding& cd, const Act* y_r, const Act* y_d, const Act& hyp_r, const Act& hyp_d, const Act& yhat_r,
                   const Act& yhat_d)
    {
        yhat_base[0] = yhat_r.p;
        yhat_base[1] = yhat_d.p;
        int c0 = 0;
        int64_t part_off = 0;
        const int h = hyp_r.h, w = hyp_r.w, HC = hyp_r.c;
        for (size_t i = 0; i < slice_ch.size(); ++i) {
            const int C = slice_ch[i];
            const size_t mark = arena.top;
            const std::string si = std::to_string(i);
            const int wide_r = 2 * C + HC + (i ? 2 * C : 0);
            const int wide_d = 4 * C + 2 * HC + (i ? 4 * C : 0);
            Act cr = alloc(hyp_r.n, h, w, wide_r), cdx = alloc(hyp_r.n, h, w, wide_d);
            copy_ch(hyp_r, view(cr, 2 * C, HC));
            copy_ch(hyp_r, view(cdx, 4 * C, HC));
            copy_ch(hyp_d, view(cdx, 4 * C + HC, HC));
            if (i) {
                const Act chr_ = view(cdx, 4 * C + 2 * HC, 2 * C), chd = view(cdx, 6 * C + 2 * HC, 2 * C);
                channel_context("rgb_channel_context." + si, view(yhat_r, 0, c0), &chr_);
                channel_context("depth_channel_context." + si, view(yhat_d, 0, c0), &chd);
                copy_ch(chr_, view(cr, 2 * C + HC, 2 * C));
            }
            const Act yr = y_r ? view(*y_r, c0, C) : Act();
            const Act yd = y_d ? view(*y_d, c0, C) : Act();
            const Act hr = view(yhat_r, c0, C), hd = view(yhat_d, c0, C);
            const int64_t part_syms = (int64_t)C * h * (w / 2);
            const Act r_loc = view(cr, 0, 2 * C), r_loc_d = view(cdx, 0, 2 * C), d_loc = view(cdx, 2 * C, 2 * C);
            // rgb anchor: [hyper_r, ch_r]
            Act p_ra = entropy_params("rgb_entropy_parameters_anchor." + si, view(cr, 2 * C, wide_r - 2 * C), 1);
            code_part(cd, 0, 1, p_ra, yr, hr, part_off);
            conv("rgb_local_context." + si, hr, 1, 2, Epi(), &r_loc);
            // depth anchor: [r_loc, hyper_r, hyper_d, ch_r, ch_d] (gathered: d_loc's slot sits in between)
            Act p_da = alloc(hyp_r.n, h, w, 2 * C);
            {
                const size_t m2 = arena.top;
                Act in = alloc(hyp_r.n, h, w, wide_d - 2 * C);
                copy_ch(r_loc, view(in, 0, 2 * C));
                copy_ch(view(cdx, 4 * C, wide_d - 4 * C), view(in, 2 * C, wide_d - 4 * C));
                entropy_params("depth_entropy_parameters_anchor." + si, in, 1, &p_da);
                arena.top = m2;
            }
            code_part(cd, 1, 1, p_da, yd, hd, part_off);
            conv("depth_local_context." + si, hd, 1, 2, Epi(), &d_loc);
            // rgb non-anchor: [r_loc, hyper_r, ch_r]
            Act p_rn = entropy_params("rgb_entropy_parameters_nonanchor." + si, cr, 2);
            code_part(cd, 0, 0, p_rn, yr, hr, part_off + part_syms);
            conv("rgb_local_context_anchor_with_nonanchor." + si, hr, 1, 2, Epi(), &r_loc_d);
            // depth non-anchor: the whole depth buffer
            Act p_dn = entropy_params("depth_entropy_parameters_nonanchor." + si, cdx, 2);
            code_part(cd, 1, 0, p_dn, yd, hd, part_off + part_syms);
            part_off += 2 * part_syms;
            c0 += C;
            arena.top = mark;
        }
    }

    // ---- single-modal ELIC (models/elic.py:15-57; BASELINE config 1) ------------------------------------------
    // analysis.py:29-52 / synthesis.py:32-70: the same blocks as above without the cross-modal fusion stages
    Act stack1(const std::string& prefix, const char* const* kinds, int n, const Act& x_in)
    {
        Act x = x_in;
        for (int i = 0; i < n; ++i) {
            const std::string k = kinds[i], name = prefix + std::to_string(i);
            if (k == "conv" || k == "deconv") x = conv(name, x, 2, 2);
            else if (k == "rb") x = bottleneck(name, x);
            else x = attention(name, x);
        }
        return x;
    }
    Act g_a1(const Act& x)
    {
        static const char* const kinds[15] = {"conv", "rb", "rb", "rb", "conv", "rb", "rb", "rb",
                                              "attn", "conv", "rb", "rb", "rb", "conv", "attn"};
        return stack1("g_a.analysis_transform.", kinds, 15, x);
    }
    Act g_s1(const Act& y)
    {
        static const char* const kinds[15] = {"attn", "deconv", "rb", "rb", "rb", "deconv", "attn", "rb",
                                              "rb", "rb", "deconv", "rb", "rb", "rb", "deconv"};
        return stack1("g_s.synthesis_transform.", kinds, 15, y);
    }
    // analysis.py:207-216
    Act h_a1(const Act& y)
    {
        Epi relu;
        relu.act = ACT_RELU;
        Act t = conv("h_a.reduction.0", y, 1, 1, relu);
        t = conv("h_a.reduction.2", t, 2, 2, relu);
        return conv("h_a.reduction.4", t, 2, 2);
    }
    // synthesis.py:276-285
    Act h_s1(const Act& zhat, const Act* dst = nullptr)
    {
        Epi relu;
        relu.act = ACT_RELU;
        Act t = conv("h_s.increase.0", zhat, 2, 2, relu);
        t = conv("h_s.increase.2", t, 2, 2, relu);
        return conv("h_s.increase.4", t, 1, 1, Epi(), dst);
    }
    // entropy.py:7-29: three 1x1 convolutions
    // `part` as in entropy_params(): 1x1 convolutions do not mix positions, so the whole net runs on one half only
    Act entropy_params1(const std::string& p, const Act& ctx, int part)
    {
        Epi relu, lin;
        relu.act = ACT_RELU;
        relu.ckbd = lin.ckbd = g_ckbd_conv ? part : 0;
        const PackedConv* last = conv_of(p + ".fusion.4.weight");
        if (!last) return Act();
        Act out = alloc(ctx.n, ctx.h, ctx.w, last->cout);
        const size_t mark = arena.top;
        Act t = conv(p + ".fusion.0", ctx, 1, 0, relu);
        t = conv(p + ".fusion.2", t, 1, 0, relu);
        conv(p + ".fusion.4", t, 1, 0, lin, &out);
        arena.top = mark;
        return out;
    }
    // elic.py:180-251 / 268-316.  Context buffer of a slice: [local 2C | channel 2C (i > 0) | hyper 2M]; the anchor
    // net reads the suffix behind the local-context slot, the non-anchor net the whole buffer.
    void bicee1(Coding& cd, const Act* y, const Act& hyper, const Act& yhat)
    {
        yhat_base[0] = yhat.p;
        int c0 = 0;
        int64_t part_off = 0;
        const int h = hyper.h, w = hyper.w, HC = hyper.c;
        for (size_t i = 0; i < slice_ch.size(); ++i) {
            const int C = slice_ch[i];
            const size_t mark = arena.top;
            const std::string si = std::to_string(i);
            const int wide = 2 * C + (i ? 2 * C : 0) + HC;
            Act ctx = alloc(hyper.n, h, w, wide);
            copy_ch(hyper, view(ctx, wide - HC, HC));
            if (i) {
                const Act cc = view(ctx, 2 * C, 2 * C);
                channel_context("channel_context." + si, view(yhat, 0, c0), &cc);
            }
            const Act ys = y ? view(*y, c0, C) : Act();
            const Act hs = view(yhat, c0, C);
            const int64_t part_syms = (int64_t)C * h * (w / 2);
            Act pa = entropy_params1("entropy_parameters_anchor." + si, view(ctx, 2 * C, wide - 2 * C), 1);
            code_part(cd, 0, 1, pa, ys, hs, part_off);
            const Act loc = view(ctx, 0, 2 * C);
            conv("local_context." + si, hs, 1, 2, Epi(), &loc);
            Act pn = entropy_params1("entropy_parameters_nonanchor." + si, ctx, 2);
            code_part(cd, 0, 0, pn, ys, hs, part_off + part_syms);
            part_off += 2 * part_syms;
            c0 += C;
            arena.top = mark;
        }
    }
    int run_compress1(const float* x_dev, int B, int H, int W, int per_image);
    int run_forward1(const float* x_dev, int B, int H, int W, float* xhat_dev, float* ly, float* lz);
    int run_decompress1(const uint8_t* const* ys, const int64_t* ylen, int n_y, const uint8_t* const* zs, const int64_t* zlen,
                        int B, int zh, int zw, float* x_out);

    // lat != nullptr: the Bi-CEE stage alone (compress_united / decompress_united): latents and hyper parameters come
    // from the caller as NCHW device tensors, the transforms and the z path are skipped
    struct Latents {
        const float* y[2];    // [B,M,h,w]     (compress only)
        const float* hyp[2];  // [B,2M,h,w]
        float* yhat[2];       // [B,M,h,w]     (decompress only)
    };
    int run_compress(const float* rgb_dev, const float* depth_dev, int B, int H, int W, int per_image,
                     const Latents* lat = nullptr);
    int run_forward(const float* rgb_dev, const float* depth_dev, int B, int H, int W, float* xr_dev, float* xd_dev,
                    float* ly_r, float* ly_d, float* lz_r, float* lz_d);
    int run_decompress_impl(const uint8_t* const* ys[2], const int64_t* ylen[2], int n_y, const uint8_t* const* zs[2],
                            const int64_t* zlen[2], int B, int h, int w, float* xr_dev, float* xd_dev, const Latents* lat);
    int run_decompress(const uint8_t* const* ys[2], const int64_t* ylen[2], int n_y, const uint8_t* const* zs[2],
                       const int64_t* zlen[2], int B, int zh, int zw, float* xr_dev, float* xd_dev);
    int ensure_arena(size_t bytes);
};

// One call through the graph cache: a dry run sizes the workspace of a new call shape, the second call of a shape captures
// its HIP graph, later ones replay it (used by every compress / decompress / forward entry point of the C ABI).
// One call = a sizing pass over the layer graph (workspace high-water mark; skipped when a cached HIP graph of this call
// shape exists, which implies the workspace already fits) and the real pass.
template <class F>
static int run_sized(rgbd_elic* m, const std::string& key, F&& run)
{
    rgbd_elic::GraphEntry* ge = m->graph_entry(key);
    if (!(ge && ge->exec)) {
        m->cur_ge = nullptr;
        m->arena.dry = true;
        m->arena.reset();
        m->arena.peak = 0;
        int r = run();
        m->arena.dry = false;
        if (r) return r;
        r = m->ensure_arena(m->arena.peak);  // a re-allocation drops every cached graph
        if (r) return r;
        ge = m->graph_entry(key);
    }
    m->cur_ge = ge;
    m->capture_failed = false;
    int r = run();
    if (r) m->body_abort();
    if (r && m->capture_failed) {
        // The capture of this call was lost before anything of its body ran.  Re-run it eagerly (the prologue is
        // idempotent); the entry tries again on its next call and retires to eager launches after kMaxCaptureFails.
        if (ge) ++ge->capture_fails;
        m->cur_ge = nullptr;
        m->capture_failed = false;
        r = run();
        if (r) m->body_abort();
        if (!r && ge) ++ge->seen;
    }
    m->cur_ge = nullptr;
    return r;
}

