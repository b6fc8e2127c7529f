// Implicit-GEMM convolution / transposed convolution for gfx950 (MI355X), fp32 in / fp32 accumulate on the
// matrix cores (v_mfma_f32_16x16x4_f32).  Replaces every nn.Conv2d / nn.ConvTranspose2d of the reference's
// ELIC_united path (modules/layers/conv.py:7-34 and the convs inside res_blk.py, attention.py, context.py,
// entropy.py, compressai/layers/layers.py) -- see DESIGN.md "K1/K2".
//
// Data layout: activations NHWC with a channel stride that is a multiple of 16 (pad channels are zero), weights
// pre-packed as [cout_pad][tap][cin_pad].  GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel] with
// k = (ci-chunk of 16, tap, ci).  One 256-thread workgroup owns TM couts x TP pixels (a TH x TW spatial tile of one
// image).  Per 16-channel chunk the input patch (tile + halo) is staged ONCE in LDS and every tap reads it at a
// shifted address, so global traffic per workgroup is patch + weights instead of taps x tile.
//
// Numerics: every output element is one k-ordered fp32 fma chain (MFMA f32 semantics) whose order depends only on
// the layer (chunk -> tap -> channel), never on tiling or batch size: results are run-to-run, batch- and
// tile-invariant, which the entropy decoder relies on to reproduce the encoder's means/scales bit for bit.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// KC = channels per LDS stage (16 or 64).  LDS rows are padded by one 16-byte slot when KC > 16 so that the 16 rows a
// ds_read_b128 lane group touches fall on different banks (row stride 272 B instead of 256 B).
template <int WM, int WN, int MT, int NT, int KC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a, int tw_log2, int tiles_x, int tiles_y,
                                                         int taps_per_stage)
{
    constexpr int TM = 16 * MT * WM;
    constexpr int TP = 16 * NT * WN;
    constexpr int RS = KC > 16 ? KC + 4 : KC;  // LDS row stride in floats
    constexpr int C4 = KC / 4;                  // 16-byte slots per row
    constexpr int WR = 8;                       // weight float4 per thread per stage (<= 32 KiB of weights per stage)
    constexpr int PR = 12;                      // patch float4 per thread per chunk (<= 48 KiB patch)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int l15 = lane & 15;
    const int q = lane >> 4;

    const int TW = 1 << tw_log2;
    const int TH = TP >> tw_log2;
    const int phase = blockIdx.z;
    const int co0 = blockIdx.y * TM;
    int bt = blockIdx.x;
    const int tile_x = bt % tiles_x;
    bt /= tiles_x;
    const int tile_y = bt % tiles_y;
    const int n = bt / tiles_y;
    const int ty0 = tile_y * TH, tx0 = tile_x * TW;

    const int PH = (TH - 1) * a.IS + a.span_y;
    const int PW = (TW - 1) * a.IS + a.span_x;
    float* patch = smem;                      // [PH*PW][RS]
    float* wl = smem + (size_t)PH * PW * RS;  // [taps_per_stage][TM][RS]

    const int ntaps = a.taps.n[phase];
    const int iy0 = ty0 * a.IS + a.min_dy;
    const int ix0 = tx0 * a.IS + a.min_dx;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane pixel coordinates of the NT column groups this wave owns (tile-local)
    int ppy[NT], ppx[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int p = (wn * NT + j) * 16 + l15;
        ppy[j] = p >> tw_log2;
        ppx[j] = p & (TW - 1);
    }

    const int npatch4 = PH * PW * C4;
    const size_t img_base = (size_t)n * a.H * a.W;
    const int nchunks = (a.cin_pad + KC - 1) / KC;
    const int ngroups = (ntaps + taps_per_stage - 1) / taps_per_stage;
    const int nstages = nchunks * ngroups;

    // Register staging.  A stage = one tap group of one channel chunk; the input patch is reloaded per chunk.
    f32x4 pw[WR], pp[PR];
    auto issue_w = [&](int stage) {
        const int ci0 = (stage / ngroups) * KC;
        const int t0 = (stage % ngroups) * taps_per_stage;
        const int nw4 = min(taps_per_stage, ntaps - t0) * TM * C4;
#pragma unroll
        for (int u = 0; u < WR; ++u) {
            const int f = tid + u * 256;
            const int c4 = f % C4;
            const int m = (f / C4) % TM;
            const int j = (f / C4) / TM;
            const int co = co0 + m;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (f < nw4 && co < a.cout_pad && ci0 + c4 * 4 < a.cin_pad)
                v = *reinterpret_cast<const f32x4*>(
                    a.w + ((size_t)co * a.ntaps_total + a.taps.wt[phase][t0 + j]) * a.cin_pad + ci0 + c4 * 4);
            pw[u] = v;
        }
    };
    auto commit_w = [&](int stage) {
        const int t0 = (stage % ngroups) * taps_per_stage;
        const int nw4 = min(taps_per_stage, ntaps - t0) * TM * C4;
#pragma unroll
        for (int u = 0; u < WR; ++u) {
            const int f = tid + u * 256;
            if (f < nw4) {
                const int c4 = f % C4;
                const int r = f / C4;  // j * TM + m
                *reinterpret_cast<f32x4*>(wl + (size_t)r * RS + c4 * 4) = pw[u];
            }
        }
    };
    auto issue_p = [&](int chunk) {
        const int ci0 = chunk * KC;
#pragma unroll
        for (int u = 0; u < PR; ++u) {
            const int f = tid + u * 256;
            const int row = f / C4, c4 = f - row * C4;
            const int pr = row / PW, pc = row - pr * PW;
            const int iy = iy0 + pr, ix = ix0 + pc;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (f < npatch4 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && ci0 + c4 * 4 < a.cin_pad)
                v = *reinterpret_cast<const f32x4*>(a.x + (img_base + (size_t)iy * a.W + ix) * a.xcs + ci0 + c4 * 4);
            pp[u] = v;
        }
    };
    auto commit_p = [&]() {
#pragma unroll
        for (int u = 0; u < PR; ++u) {
            const int f = tid + u * 256;
            if (f < npatch4) {
                const int row = f / C4, c4 = f - row * C4;
                *reinterpret_cast<f32x4*>(patch + (size_t)row * RS + c4 * 4) = pp[u];
            }
        }
    };

    for (int stage = 0; stage < nstages; ++stage) {
        const int t0 = (stage % ngroups) * taps_per_stage;
        const int tg = min(taps_per_stage, ntaps - t0);
        // all global loads of the stage are issued back to back (one memory round trip per stage), then committed to
        // LDS; the staging registers are dead during the MFMA phase so two workgroups fit per CU and hide each other's
        // load phase
        issue_w(stage);
        if (t0 == 0) issue_p(stage / ngroups);
        __syncthreads();  // every wave has finished reading the previous stage from LDS
        if (t0 == 0) commit_p();
        commit_w(stage);
        __syncthreads();
        // canonical accumulation order: 16-channel chunk -> tap -> channel.  A 64-channel stage therefore walks its
        // four sub-chunks in the OUTER loop (the launcher only picks KC=64 when one stage holds every tap), so the
        // fma chain of each output is the same for every KC / tile choice.
#pragma unroll
        for (int kk = 0; kk < KC / 16; ++kk) {
            for (int j = 0; j < tg; ++j) {
                const int dy = a.taps.dy[phase][t0 + j] - a.min_dy;
                const int dx = a.taps.dx[phase][t0 + j] - a.min_dx;
                f32x4 af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[i] = *reinterpret_cast<const f32x4*>(
                        wl + ((size_t)j * TM + (wm * MT + i) * 16 + l15) * RS + kk * 16 + q * 4);
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const int row = (ppy[k] * a.IS + dy) * PW + ppx[k] * a.IS + dx;
                    bf[k] = *reinterpret_cast<const f32x4*>(patch + (size_t)row * RS + kk * 16 + q * 4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int k = 0; k < NT; ++k)
                            acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[k][e], acc[i][k], 0, 0, 0);
            }
        }
    }

    // epilogue: lane holds couts cb..cb+3 of pixel (ppy,ppx) for each (i,k)
    const int oy_off = a.nphase > 1 ? (phase >> 1) : 0;
    const int ox_off = a.nphase > 1 ? (phase & 1) : 0;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const int gy = ty0 + ppy[k], gx = tx0 + ppx[k];
        if (gy >= a.GH || gx >= a.GW) continue;
        const int oy = gy * a.OS + oy_off, ox = gx * a.OS + ox_off;
        const size_t pix = ((size_t)n * a.OH + oy) * a.OW + ox;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int cb = co0 + (wm * MT + i) * 16 + q * 4;
            if (cb >= a.cout_pad) continue;
            f32x4 v = acc[i][k];
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + cb);
            v += b;
            if (a.res1) v += *reinterpret_cast<const f32x4*>(a.res1 + pix * a.r1cs + cb);
            if (a.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (a.act == ACT_LEAKY) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * 0.01f;
            } else if (a.act == ACT_SIGMOID) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + expf(-v[e]));
            }
            if (a.mul) v *= *reinterpret_cast<const f32x4*>(a.mul + pix * a.mcs + cb);
            if (a.res2) v += *reinterpret_cast<const f32x4*>(a.res2 + pix * a.r2cs + cb);
            *reinterpret_cast<f32x4*>(a.y + pix * a.ycs + cb) = v;
        }
    }
}

namespace {

constexpr int LDS_BUDGET = 78 * 1024;  // two workgroups per CU (160 KiB LDS)

template <int WM, int WN, int MT, int NT, int KC>
int launch_cfg(const ConvArgs& a, int tw_log2, hipStream_t s)
{
    constexpr int TM = 16 * MT * WM;
    constexpr int TP = 16 * NT * WN;
    constexpr int RS = KC > 16 ? KC + 4 : KC;
    const int TW = 1 << tw_log2, TH = TP / TW;
    const int tiles_x = (a.GW + TW - 1) / TW, tiles_y = (a.GH + TH - 1) / TH;
    const int PH = (TH - 1) * a.IS + a.span_y, PW = (TW - 1) * a.IS + a.span_x;
    const size_t patch_bytes = (size_t)PH * PW * RS * sizeof(float);
    const size_t tap_bytes = (size_t)TM * RS * sizeof(float);
    int max_taps = 1;
    for (int p = 0; p < a.nphase; ++p) max_taps = a.taps.n[p] > max_taps ? a.taps.n[p] : max_taps;
    if ((long)PH * PW * (KC / 4) > 12 * 256) return RGBD_ENOSPC;  // patch registers (PR)
    long room = (8 * 256) / ((long)TM * (KC / 4));               // weight registers (WR)
    if (room < 1) return RGBD_ENOSPC;
    const int tps = (int)(room < max_taps ? room : max_taps);
    if (KC > 16 && tps < max_taps) return RGBD_ENOSPC;  // would break the canonical accumulation order
    const size_t lds = patch_bytes + (size_t)tps * tap_bytes;
    auto kern = conv_mfma_kernel<WM, WN, MT, NT, KC>;
    static size_t configured = 0;  // per instantiation
    if (lds > 64 * 1024 && lds > configured) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(160 * 1024)));
        configured = 160 * 1024;
    }
    dim3 grid((unsigned)(tiles_x * tiles_y * a.N), (unsigned)((a.cout_pad + TM - 1) / TM), (unsigned)a.nphase);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, tw_log2, tiles_x, tiles_y, tps);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// tile-width choice for a TP-pixel tile: avoid ragged last tiles on narrow feature maps
int pick_tw_log2(int GW, int GH, int TP)
{
    int best = 4;
    long best_cost = -1;
    for (int l = 2; l <= 4; ++l) {
        const int TW = 1 << l, TH = TP / TW;
        if (TH < 1) continue;
        const long cost = (long)((GW + TW - 1) / TW) * ((GH + TH - 1) / TH);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && l > best)) {
            best = l;
            best_cost = cost;
        }
    }
    return best;
}

struct Choice {
    int wm, mt, nt, kc, tw_log2;
};

Choice choose(const ConvArgs& a)
{
    // Pick the tile by a small cost model (cycles per CU): MFMA time of the workgroups a CU has to run vs the bytes
    // each workgroup pulls from L2 (its weight slab K*TM plus the input patch per channel chunk).  Multi-tap layers on
    // small feature maps are weight-traffic bound, so narrow cout tiles x wide pixel tiles win there; big feature maps
    // pick the large square-ish tiles.  The choice never changes results: every output keeps its fma chain.
    static const int cand[][3] = {{2, 5, 4}, {2, 4, 4}, {2, 3, 4}, {2, 2, 4}, {2, 1, 4}, {2, 5, 2}, {2, 4, 2},
                                  {2, 3, 2}, {2, 2, 2}, {2, 1, 2}, {2, 5, 1}, {2, 4, 1}, {2, 3, 1}, {2, 2, 1},
                                  {2, 1, 1}, {1, 3, 2}, {1, 2, 2}, {1, 1, 2}, {1, 3, 1}, {1, 2, 1}, {1, 1, 1}};
    int max_taps = 1;
    for (int p = 0; p < a.nphase; ++p) max_taps = a.taps.n[p] > max_taps ? a.taps.n[p] : max_taps;
    const double K = (double)max_taps * a.cin_pad;
    Choice best{2, 4, 4, 16, 4};
    double best_cost = -1.0;
    for (const auto& c : cand) {
        const int wm = c[0], mt = c[1], nt = c[2];
        const int tm = 16 * mt * wm, tp = 16 * nt * (wm == 2 ? 2 : 4);
        const int twl = pick_tw_log2(a.GW, a.GH, tp);
        const int TW = 1 << twl, TH = tp / TW;
        const long tiles = (long)((a.GW + TW - 1) / TW) * ((a.GH + TH - 1) / TH) * a.N;
        const long blocks = tiles * ((a.cout_pad + tm - 1) / tm) * a.nphase;
        const int PH = (TH - 1) * a.IS + a.span_y, PW = (TW - 1) * a.IS + a.span_x;
        if ((long)PH * PW * 4 > 12 * 256 || (long)tm * 4 > 8 * 256) continue;  // register-staging limits (KC=16)
        const double mfma = (double)mt * nt * (K / 4.0) * 32.0 + 3000.0;             // cycles per workgroup (+ prologue)
        const double load = (K * tm * 4.0 + (double)PH * PW * a.cin_pad * 4.0) / 12.0;  // ~12 B/clk/CU from L2
        const long nb = (blocks + 255) / 256;
        const double cost = nb >= 2 ? nb * (mfma > load ? mfma : load) * (blocks < 512 ? 1.15 : 1.0) : (mfma + load);
        if (best_cost < 0 || cost < best_cost * 0.999) {
            best_cost = cost;
            best = Choice{wm, mt, nt, 16, twl};
        }
    }
    // channels per stage: 64 when one stage can hold every tap (keeps the canonical order) and a 16-channel stage
    // would hold too few MFMAs between barriers
    if (a.cin_pad >= 64) {
        const int tp = 16 * best.nt * (best.wm == 2 ? 2 : 4);
        const int TW = 1 << best.tw_log2, TH = tp / TW;
        const size_t patch64 = (size_t)((TH - 1) * a.IS + a.span_y) * ((TW - 1) * a.IS + a.span_x) * 68 * 4;
        const size_t tap64 = (size_t)16 * best.mt * best.wm * 68 * 4;
        const long mfma_per_stage16 = (long)max_taps * 4 * best.mt * best.nt;
        const long p4 = (long)((TH - 1) * a.IS + a.span_y) * ((TW - 1) * a.IS + a.span_x) * 16;
        const long w4 = (long)max_taps * 16 * best.mt * best.wm * 16;
        if (patch64 + (size_t)max_taps * tap64 <= (size_t)LDS_BUDGET && mfma_per_stage16 < 400 && p4 <= 12 * 256 &&
            w4 <= 8 * 256)
            best.kc = 64;
    }
    return best;
}

#define RGBD_CASE(WM_, WN_, MT_, NT_)                                              \
    if (c.wm == WM_ && c.mt == MT_ && c.nt == NT_)                                 \
        return c.kc == 64 ? launch_cfg<WM_, WN_, MT_, NT_, 64>(a, c.tw_log2, s)    \
                          : launch_cfg<WM_, WN_, MT_, NT_, 16>(a, c.tw_log2, s);

}  // namespace

int launch_conv(const ConvArgs& a, hipStream_t s)
{
    if (a.cin_pad % 16 || a.cout_pad % 16 || a.xcs % 4 || a.ycs % 4) return RGBD_EINVAL;
    if (a.nphase != 1 && a.nphase != 4) return RGBD_EINVAL;
    if (a.N <= 0 || a.GH <= 0 || a.GW <= 0) return RGBD_EINVAL;
    const Choice c = choose(a);
    RGBD_CASE(2, 2, 5, 4) RGBD_CASE(2, 2, 4, 4) RGBD_CASE(2, 2, 3, 4) RGBD_CASE(2, 2, 2, 4) RGBD_CASE(2, 2, 1, 4)
    RGBD_CASE(2, 2, 5, 2) RGBD_CASE(2, 2, 4, 2) RGBD_CASE(2, 2, 3, 2) RGBD_CASE(2, 2, 2, 2) RGBD_CASE(2, 2, 1, 2)
    RGBD_CASE(2, 2, 5, 1) RGBD_CASE(2, 2, 4, 1) RGBD_CASE(2, 2, 3, 1) RGBD_CASE(2, 2, 2, 1) RGBD_CASE(2, 2, 1, 1)
    RGBD_CASE(1, 4, 3, 2) RGBD_CASE(1, 4, 2, 2) RGBD_CASE(1, 4, 1, 2)
    RGBD_CASE(1, 4, 3, 1) RGBD_CASE(1, 4, 2, 1) RGBD_CASE(1, 4, 1, 1)
    return RGBD_EINVAL;
}
