// Shared kernel templates of the convolution launchers (conv_mfma.hip: the k-ordered single-chain kernels; conv_mfma_blk.hip:
// the blocked-accumulation kernels).  See conv_mfma.hip for the layout and numerics notes.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>

#include "common.h"
#include "exact_math.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// KC = channels per LDS stage (16 or 64).  LDS rows are padded by one 16-byte slot when KC > 16 so that the 16 rows a
// ds_read_b128 lane group touches fall on different banks (row stride 272 B instead of 256 B).
// DMA (KC == 16 only): operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers) into a
// double-buffered LDS image; the loads of stage s+1 are in flight under the MFMAs of stage s and one barrier per stage
// both retires them (vmcnt) and frees the other buffer.  Out-of-image patch slots are zeroed with ordinary LDS stores.
// Without DMA the stage is staged through registers (all loads issued back to back, then committed).
// 1 / (1 + e^-x) with the hardware exp2 / rcp (1 ulp each): four instructions per value.  The libm expf and the IEEE
// division expand to ~60 instructions per value, unrolled for every output slot of the epilogue -- together with GELU's
// erff that was three quarters of the kernel's code (8150 -> 2100 instructions per instantiation).
__device__ __forceinline__ float sigmoid_f32(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// NB > 2 (DMA, single-tap stride-1 layers only): a ring of NB stage buffers instead of the double buffer.  A 1x1 layer has
// MT*NT*4 MFMAs per 16-channel stage (0.2 - 0.9 us) -- less than one global round trip, so with one stage in flight the
// wave waits for memory at every stage and a CU never has more than one (TM + TP) x 64-byte stage per workgroup on the way
// (measured: ~2.7 TB/s over the chip whatever the tile).  The ring keeps NB - 1 stages in flight: stage s's
// loads are waited for with s_waitcnt vmcnt(<loads of the NB - 2 stages issued after it>), not vmcnt(0).
template <int N>
__device__ __forceinline__ void rgbd_wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BLK (conv_mfma_blk.hip): blocked accumulation.  The reduction is cut into blocks that end after the 16-channel chunks
// marked in ConvArgs::blk_end; every block is its own fma chain that starts from zero (`acc`), and the finished block sums
// are added one after the other to a running total (`tot`): total_{j+1} = fl(total_j + S_j).  That is the arithmetic of the
// CPU library the reference runs on (oneDNN's jit:avx512_core kernels keep one accumulator per 16-channel block and add it to
// the destination) -- and a shorter error chain than one k-ordered sum.  ConvArgs::bias_mode says where the
// bias enters: 0 the epilogue (after the sum), 1 the running total starts from it (total = S_0 + bias), 2 the first chain
// starts from it.
template <int WM, int WN, int MT, int NT, int KC, bool DMA, int G2, bool LEAD = false, int NB = 2, bool BLK = false>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs& a, int tw_log2, int tiles_x, int tiles_y, int taps_per_stage,
                                               int tab_f)
{
    constexpr int TM = 16 * MT * WM;
    constexpr int TP = 16 * NT * WN;
    constexpr bool RING = NB > 2;
    static_assert(!RING || (DMA && KC == 16 && G2 == 0 && TP % 64 == 0), "ring staging: DMA, 16-channel stages, whole waves of patch slots");
    constexpr int RS = KC > 16 ? KC + 4 : KC;  // LDS row stride in floats
    constexpr int C4 = KC / 4;                  // 16-byte slots per row
    constexpr int WR = 8;                       // weight float4 per thread per stage (<= 32 KiB of weights per stage)
    constexpr int PR = TP >= 128 ? 12 : (TP >= 64 ? 6 : 4);  // patch float4 per thread per chunk
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
    const int phase = blockIdx.z / a.splitk;
    const int split = blockIdx.z - phase * a.splitk;
    // Workgroup -> (pixel tile, cout tile).  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so
    // the linear id is first remapped to give every XCD a contiguous chunk of the work list (bijective for any count),
    // and the work list runs cout-tile-fastest: the workgroups that read the same input patch then sit on one XCD back
    // to back and the second one finds the patch in that L2 instead of HBM.  Placement is a speed matter only.
    const int ycount = (a.cout_pad + TM - 1) / TM;
    int wid;
    {
        const int orig = blockIdx.x, nwg = gridDim.x;
        const int qd = nwg >> 3, rm = nwg & 7, xcd = orig & 7;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
    }
    const int co0 = (wid % ycount) * TM;
    int bt = wid / ycount;
    const int tile_x = bt % tiles_x;
    bt /= tiles_x;
    const int tile_y = bt % tiles_y;
    int n = bt / tiles_y;
    // grouped launch (ConvArgs::groups == 2): the second half of the image range runs the same layer on the second operand
    // set.  Everything below addresses through these wave-uniform pointers (scalar selects on kernel arguments).
    const bool g1 = n >= a.N;
    if (g1) n -= a.N;
    const float* const gx = g1 ? a.g1.x : a.x;
    const float* const gw = g1 ? a.g1.w : a.w;
    const float* const gbias = g1 ? a.g1.bias : a.bias;
    float* const gy = g1 ? a.g1.y : a.y;
    const float* const gres1 = g1 ? a.g1.res1 : a.res1;
    const float* const gmul = g1 ? a.g1.mul : a.mul;
    const float* const gres2 = g1 ? a.g1.res2 : a.res2;
    float* const gpartial = g1 ? a.g1.partial : a.partial;
    float* const gy2 = g1 ? a.g1.y2 : a.y2;
    const float* const gw2 = g1 ? a.g1.w2 : a.w2;
    const float* const gbias2 = g1 ? a.g1.bias2 : a.bias2;
    const float* const gw3 = g1 ? a.g1.w3 : a.w3;
    const float* const gbias3 = g1 ? a.g1.bias3 : a.bias3;
    float* const gy3 = g1 ? a.g1.y3 : a.y3;
    // checkerboard output (a.ckbd): the tile's TW columns are every second column of a 2*TW-wide strip -- pixel (py, k)
    // sits at column 2k + par(row); the staged patch is the whole strip, only the B-fragment rows and the stores move
    const int ck = a.ckbd ? 1 : 0;
    const int ty0 = tile_y * TH, tx0 = tile_x * (TW << ck);

    const int PH = (TH - 1) * a.IS + a.span_y;
    const int PW = ((TW << ck) - 1) * a.IS + a.span_x;
    const int patch_f = PH * PW * RS;                         // floats per patch buffer
    const int wl_f = taps_per_stage * TM * RS;                 // floats per weight buffer
    constexpr int NBUF = DMA ? NB : 1;
    float* patch = smem;                                       // [NBUF][PH*PW][RS]
    float* wl = smem + (size_t)patch_f * NBUF;                 // [NBUF][taps_per_stage][TM][RS]

    const int ntaps = a.taps.n[phase];
    // The tap table is int8 data in the kernel-argument segment; indexing it in the loops below would be a *vector* global
    // load per tap (there are no sub-dword scalar loads) sitting in front of every MFMA burst and every weight fetch.
    // Each workgroup therefore expands its phase's taps once into LDS (behind the stage / epilogue buffers):
    //   tap_off[t] = LDS float offset of tap t inside the patch, tap_w[t] = its weight-slab index
    int* tap_off = reinterpret_cast<int*>(smem + tab_f);
    int* tap_w = tap_off + 32;
    if (tid < ntaps) {
        tap_off[tid] = ((a.taps.dy[phase][tid] - a.min_dy) * PW + (a.taps.dx[phase][tid] - a.min_dx)) * RS;
        tap_w[tid] = (int)a.taps.wt[phase][tid] * a.cin_pad;
    }
    __syncthreads();
    const int my_tap_off = tap_off[lane & 31];  // lane t keeps tap t's offset: the loops fetch it with v_readlane
    const int my_tap_w = tap_w[lane & 31];      // likewise its weight-slab offset
    const int iy0 = ty0 * a.IS + a.min_dy;
    const int ix0 = tx0 * a.IS + a.min_dx;

    f32x4 acc[MT][NT];
    f32x4 tot[BLK ? MT : 1][BLK ? NT : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        // (the accumulator registers of cout tile i hold couts co0 + (wm*MT + i)*16 + 4q .. +3 of this lane's pixels)
        f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int cob = co0 + (wm * MT + i) * 16 + q * 4;
        if (a.bias_mode != 0 && cob < a.cout_pad) b4 = *reinterpret_cast<const f32x4*>(gbias + cob);
        const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[i][j] = (a.bias_mode == 2 && split == 0) ? b4 : z4;
            if constexpr (BLK) tot[i][j] = a.bias_mode == 1 ? b4 : z4;
        }
    }
    // BLK: the chain that ends with 16-channel chunk `c16` of the layer is added to the running total when the layer's block
    // table says so (wave-uniform: scalar loads of kernel arguments)
    auto fold = [&](int c16) {
        if constexpr (BLK) {
            if ((a.blk_end[c16 >> 5] >> (c16 & 31)) & 1u) {
                // The block sums are read by vector instructions right behind the MFMAs that produce them, across a loop exit --
                // where this compiler's hazard recogniser was seen to leave out the wait states an 8-pass MFMA result needs on
                // gfx950 (12; one result tile per wave: the tile's last lane group was read stale, 10 instructions behind its
                // MFMA).  Earlier tiles are covered by the MFMAs issued after theirs (8 wait states each); the last two are tied
                // to an explicit wait here (conv_mfma_blk.hip is built with MFMA results in VGPRs, so "+v" moves nothing).
                auto tie = [](f32x4& t, bool wait) {
                    float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
                    if (wait) asm volatile("s_nop 11" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                    else asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                    t = (f32x4){t0, t1, t2, t3};
                };
                tie(acc[MT - 1][NT - 1], true);
                if constexpr (NT > 1) tie(acc[MT - 1][NT - 2], false);
                else if constexpr (MT > 1) tie(acc[MT - 2][NT - 1], false);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        tot[i][j] += acc[i][j];
                        acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
            }
        }
    };

    // per-lane pixel coordinates of the NT column groups this wave owns (tile-local)
    int ppy[NT], ppx[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int p = (wn * NT + j) * 16 + l15;
        ppy[j] = p >> tw_log2;
        ppx[j] = p & (TW - 1);
        if (ck) ppx[j] = 2 * ppx[j] + (((ty0 + ppy[j]) & 1) ^ (a.ckbd == 1 ? 1 : 0));
    }

    const int npatch4 = PH * PW * C4;
    int brow0[NT];  // LDS float offset of this lane's pixel (tap (min_dy, min_dx)) for each column group
#pragma unroll
    for (int j = 0; j < NT; ++j) brow0[j] = ((ppy[j] * a.IS) * PW + ppx[j] * a.IS) * RS + q * 4;
    // split-K: this workgroup reduces 16-channel chunks [c16_lo, c16_hi) only (the split is a fixed function of the
    // layer, so every output keeps one well-defined summation order: chain per split, then splits in order)
    const int n16 = a.cin_pad / 16;
    const int per = (n16 + a.splitk - 1) / a.splitk;
    // (ConvArgs::split_c16, when set: the ranges are the layer's accumulation blocks -- each split is one block's chain and the
    //  reducer adds the block sums in order, which is the blocked sum of conv_mfma_blk.hip spread over workgroups)
    const bool sb = a.split_c16[a.splitk] != 0;
    const int c16_lo = sb ? a.split_c16[split] : split * per, c16_hi = sb ? a.split_c16[split + 1] : min(n16, c16_lo + per);
    const int ci_lo = c16_lo * 16, ci_hi = c16_hi * 16;
    const int nchunks = (ci_hi - ci_lo + KC - 1) / KC;
    const int ngroups = (ntaps + taps_per_stage - 1) / taps_per_stage;
    const int nstages = nchunks > 0 ? nchunks * ngroups : 0;

    // Register staging.  A stage = one tap group of one channel chunk; the input patch is reloaded per chunk.
    // Everything that does not change from stage to stage is computed once here: per staging slot u the thread's
    // global element offsets (weights: without the tap/chunk term; patch: pixel offset or -1 outside the image) --
    // the per-stage work is then one add and one 16-byte load per slot.  LDS destinations are affine in u.
    f32x4 pw[DMA ? 1 : WR], pp[DMA ? 1 : PR];
    int gw_j[WR];      // tap slot inside the stage
    unsigned gw_voff[WR];  // byte offset of the slot's weight row ((co0+m) * ntaps_total * cin_pad + c4*4 floats; always a valid row)
    int gp_off[PR];    // (iy*W + ix) * xcs + c4*4 inside image n, or -1 when outside the image / patch
    const float* xn = gx + (size_t)n * a.H * a.W * a.xcs;  // wave-uniform base of the tile's image
    {
#pragma unroll
        for (int u = 0; u < WR; ++u) {
            const int f = tid + u * 256;
            const int m = (f / C4) % TM;
            int c4 = f % C4;
            // ring staging: the LDS image is XOR-swizzled (see the fragment reads of the ring loop): slot s of row m holds the
            // channel quad s ^ ((-(m >> 2)) & 3)
            if constexpr (RING) c4 ^= (4 - ((m >> 2) & 3)) & 3;
            gw_j[u] = __builtin_amdgcn_readfirstlane((f / C4) / TM);  // a wave's 64 slots are 64/C4 rows of one tap (TM % 16 == 0)
            const int co = co0 + m;
            // rows past cout_pad re-read the tile's first row instead of being masked off (their outputs are never
            // stored), so the loads need no per-lane predicate
            gw_voff[u] = (unsigned)((co < a.cout_pad ? co : co0) * a.ntaps_total * a.cin_pad + c4 * 4) * 4u;
        }
#pragma unroll
        for (int u = 0; u < PR; ++u) {
            const int f = tid + u * 256;
            const int row = f / C4;
            int c4 = f - row * C4;
            if constexpr (RING) c4 ^= (4 - ((row >> 2) & 3)) & 3;  // (the swizzled image, as for the weights)
            const int pr = row / PW, pc = row - pr * PW;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = f < npatch4 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            gp_off[u] = ok ? (iy * a.W + ix) * a.xcs + c4 * 4 : -1;  // inside image n (xn below): fits 32 bits
            if constexpr (RING)  // single tap, no halo: a slot outside the image belongs to a pixel that is never stored -- it
                                 // re-reads a valid pixel so that every wave issues the same number of loads per stage
                gp_off[u] = (min(max(iy, 0), a.H - 1) * a.W + min(max(ix, 0), a.W - 1)) * a.xcs + c4 * 4;
        }
    }
    const int w_lds0 = (tid / C4) * RS + (tid % C4) * 4;  // slot u adds u * (256 / C4) * RS floats
    auto issue_w = [&](int stage) {
        const int ci0 = ci_lo + (stage / ngroups) * KC;
        const int t0 = (stage % ngroups) * taps_per_stage;
        const int tg = min(taps_per_stage, ntaps - t0);
#pragma unroll
        for (int u = 0; u < WR; ++u) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int c4x4 = ((tid + u * 256) % C4) * 4;
            if (gw_j[u] < tg && (KC == 16 || ci0 + c4x4 < ci_hi))
                v = *reinterpret_cast<const f32x4*>(
                    reinterpret_cast<const char*>(gw) +
                    (size_t)(gw_voff[u] + (unsigned)(__builtin_amdgcn_readlane(my_tap_w, t0 + gw_j[u]) + ci0) * 4u));
            pw[DMA ? 0 : u] = v;
        }
    };
    auto commit_w = [&](int stage) {
        const int t0 = (stage % ngroups) * taps_per_stage;
        const int nrows = min(taps_per_stage, ntaps - t0) * TM;
#pragma unroll
        for (int u = 0; u < WR; ++u)
            if (tid / C4 + u * (256 / C4) < nrows)
                *reinterpret_cast<f32x4*>(wl + w_lds0 + u * (256 / C4) * RS) = pw[DMA ? 0 : u];
    };
    auto issue_p = [&](int chunk) {
        const int ci0 = ci_lo + chunk * KC;
#pragma unroll
        for (int u = 0; u < PR; ++u) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int c4x4 = ((tid + u * 256) % C4) * 4;
            if (gp_off[u] >= 0 && (KC == 16 || ci0 + c4x4 < ci_hi))
                v = *reinterpret_cast<const f32x4*>(xn + gp_off[u] + ci0);
            pp[DMA ? 0 : u] = v;
        }
    };
    auto commit_p = [&]() {
#pragma unroll
        for (int u = 0; u < PR; ++u)
            if (tid + u * 256 < npatch4) *reinterpret_cast<f32x4*>(patch + w_lds0 + u * (256 / C4) * RS) = pp[DMA ? 0 : u];
    };
    // DMA variants: LDS slot f = tid + u*256 is 16 bytes at f*16 (RS == 16: the image is lane-linear), so a wave's
    // destination base is uniform and lane i lands at base + 16*i
    const int wave_slot0 = (__builtin_amdgcn_readfirstlane(tid) >> 6) * 64;
    auto dma_w = [&](int stage, int buf) {
        const int ci0 = ci_lo + (stage / ngroups) * KC;
        const int t0 = (stage % ngroups) * taps_per_stage;
        const int tg = min(taps_per_stage, ntaps - t0);
        float* lds_w = wl + buf * wl_f + wave_slot0 * 4;
        // address = wave-uniform pointer (tap slab + channel chunk, SALU) + hoisted per-lane byte offset: a handful of
        // scalar instructions per load and no exec masking -- the issue slots of this code come out of the MFMA stream
#pragma unroll
        for (int u = 0; u < WR; ++u)
            if (gw_j[u] < tg) {  // wave-uniform
                const unsigned so = (unsigned)(__builtin_amdgcn_readlane(my_tap_w, t0 + gw_j[u]) + ci0) * 4u;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(gw) + (size_t)(gw_voff[u] + so)),
                    (__attribute__((address_space(3))) void*)(lds_w + u * 1024), 16, 0, 0);
            }
    };
    auto dma_p = [&](int chunk, int buf) {
        const int ci0 = ci_lo + chunk * KC;
#pragma unroll
        for (int u = 0; u < PR; ++u) {
            const int f = tid + u * 256;
            if (f < npatch4) {
                if (gp_off[u] >= 0)  // uniform base + 32-bit per-lane byte offset (saddr form, no 64-bit VALU adds)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(xn) +
                                                                        (size_t)((unsigned)(gp_off[u] + ci0) * 4u)),
                        (__attribute__((address_space(3))) void*)(patch + buf * patch_f + (wave_slot0 + u * 256) * 4), 16, 0, 0);
                else
                    *reinterpret_cast<f32x4*>(patch + buf * patch_f + f * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // epilogue geometry (see below)
    constexpr int EMT = (TP * (16 * WM * MT + 4) * 4 <= 64 * 1024) ? MT : (MT + 1) / 2;  // cout tiles per pass
    constexpr int SW = 16 * WM * EMT + 4;                                                   // staging row stride (floats)
    constexpr int S4 = 4 * WM * EMT;                                                        // float4 per staged pixel row
    constexpr int EU = TP * S4 / 256;
    const int oy_off = a.nphase > 1 ? (phase >> 1) : 0;
    const int ox_off = a.nphase > 1 ? (phase & 1) : 0;
    if constexpr (RING) {
        // loads per stage of this wave: patch TP*4/256 each; weights (TM*4 + 255)/256 for the first waves, one fewer for the rest
        constexpr int PL = TP * 4 / 256;
        constexpr int WLO = TM * 4 / 256, WREM = (TM * 4 % 256) / 64;  // waves [0, WREM) issue WLO + 1 weight loads
        const bool more = wave < WREM;
        static_assert(PL <= PR && WLO + 1 <= WR, "ring staging slots");
        const unsigned tapw0 = (unsigned)__builtin_amdgcn_readlane(my_tap_w, 0);
        const int nst = __builtin_amdgcn_readfirstlane(nstages);  // (wave-uniform; keeps the loop control on the scalar unit)
        // one stage = this wave's PL patch loads + WLO (+1) weight loads: no predicates, no per-lane branches
        auto ring_issue = [&](int st, int buf) {
            const unsigned ci0 = (unsigned)(ci_lo + st * 16);
            float* lds_p = patch + buf * patch_f + wave_slot0 * 4;
            float* lds_w = wl + buf * wl_f + wave_slot0 * 4;
#pragma unroll
            for (int u = 0; u < PL; ++u)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(xn) + (size_t)(((unsigned)gp_off[u] + ci0) * 4u)),
                    (__attribute__((address_space(3))) void*)(lds_p + u * 1024), 16, 0, 0);
#pragma unroll
            for (int u = 0; u < WLO; ++u)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(gw) + (size_t)(gw_voff[u] + (tapw0 + ci0) * 4u)),
                    (__attribute__((address_space(3))) void*)(lds_w + u * 1024), 16, 0, 0);
            if (more)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(gw) + (size_t)(gw_voff[WLO] + (tapw0 + ci0) * 4u)),
                    (__attribute__((address_space(3))) void*)(lds_w + WLO * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int st = 0; st < NB - 1; ++st)
            if (st < nst) ring_issue(st, st);
        for (int stage = 0; stage < nst; ++stage) {
            // stage's loads have landed once at most the loads of the NB - 2 stages issued after it are outstanding
            if (stage + NB - 2 < nst) {
                if (more) rgbd_wait_vmcnt<(NB - 2) * (PL + WLO + 1)>();
                else rgbd_wait_vmcnt<(NB - 2) * (PL + WLO)>();
            } else {
                rgbd_wait_vmcnt<0>();  // tail: fewer stages behind this one
            }
            asm volatile("s_barrier" ::: "memory");  // everyone's part of the stage is in LDS; everyone has left stage - 1
            if (stage + NB - 1 < nst) ring_issue(stage + NB - 1, (stage + NB - 1) % NB);
            const int buf = stage % NB;
            const float* cur_w = wl + buf * wl_f;
            const float* cur_p = patch + buf * patch_f;
            // Fragment reads from the swizzled image.  A ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15,
            // 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) -- one LDS cycle per group when its 16 addresses
            // fall into 16 different bank quads (address / 16 mod 16).  With plain 64-byte rows, lane (row l15, quad q) reads
            // address (16 base + l15) * 64 + 16 q: rows l15 and l15 + 4 k share a bank quad, every group is a 2-way conflict
            // and the read takes 8 array cycles instead of 4 (a 1x1 stage is 7 such reads per 48 MFMAs and wave).  Slot
            // q ^ ((-(l15 >> 2)) & 3) of the row instead: the four (q, l15 >> 2) pairs of each group land on four different
            // slot columns -- conflict-free -- and the direct-to-LDS loads put channel quad q there for free (the lane that
            // fills slot s of row m simply loads quad s ^ ((-(m >> 2)) & 3)).
            const int qs = (q ^ ((4 - (l15 >> 2)) & 3)) * 4;
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(cur_w + ((wm * MT + i) * 16 + l15) * RS + qs);
#pragma unroll
            for (int k = 0; k < NT; ++k) bf[k] = *reinterpret_cast<const f32x4*>(cur_p + brow0[k] - q * 4 + qs);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int k = 0; k < NT; ++k)
                        acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[k][e], acc[i][k], 0, 0, 0);
            fold(c16_lo + stage);
        }
    } else {
    if (DMA && nstages > 0) {
        dma_p(0, 0);
        dma_w(0, 0);
    }
    for (int stage = 0; stage < nstages; ++stage) {
        const int t0 = (stage % ngroups) * taps_per_stage;
        const int tg = min(taps_per_stage, ntaps - t0);
        const float* cur_w = wl;
        const float* cur_p = patch;
        if (DMA) {
            const int buf = stage & 1, pbuf = (stage / ngroups) & 1;
            __syncthreads();  // retires this stage's DMA (vmcnt) and frees the other buffers (everyone left stage-1)
            if (stage + 1 < nstages) {
                dma_w(stage + 1, buf ^ 1);
                if ((stage + 1) % ngroups == 0) dma_p((stage + 1) / ngroups, pbuf ^ 1);
            }
            cur_w = wl + buf * wl_f;
            cur_p = patch + pbuf * patch_f;
        } else {
            // all global loads of the stage are issued back to back (one memory round trip per stage), then committed
            // to LDS; the staging registers are dead during the MFMA phase so two workgroups fit per CU and hide each
            // other's load phase (keeping the next stage's loads in flight during the MFMAs was measured for the
            // small-accumulator variants that have the registers for it: no gain)
            issue_w(stage);
            if (t0 == 0) issue_p(stage / ngroups);
            __syncthreads();  // every wave has finished reading the previous stage from LDS
            if (t0 == 0) commit_p();
            commit_w(stage);
            __syncthreads();
        }
        // canonical accumulation order: 16-channel chunk -> tap -> channel.  A 64-channel stage therefore walks its
        // four sub-chunks in the OUTER loop (the launcher only picks KC=64 when one stage holds every tap), so the
        // fma chain of each output is the same for every KC / tile choice.
        const int nkk = min(KC / 16, (ci_hi - ci_lo - (stage / ngroups) * KC) / 16);  // no MFMAs on the zero tail
        // (double-buffering the A/B fragments across taps was measured: +60 VGPRs drop the 128-pixel tiles to one
        //  workgroup per CU and cost more than the hidden LDS latency gains)
        for (int kk = 0; kk < nkk; ++kk) {
            for (int j = 0; j < tg; ++j) {
                const int toff = __builtin_amdgcn_readlane(my_tap_off, t0 + j) + kk * 16;  // wave-uniform
                f32x4 af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[i] = *reinterpret_cast<const f32x4*>(cur_w + (j * TM + (wm * MT + i) * 16 + l15) * RS + kk * 16 + q * 4);
#pragma unroll
                for (int k = 0; k < NT; ++k) bf[k] = *reinterpret_cast<const f32x4*>(cur_p + brow0[k] + toff);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int k = 0; k < NT; ++k)
                            acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[k][e], acc[i][k], 0, 0, 0);
            }
            if (t0 + tg == ntaps) fold(c16_lo + (stage / ngroups) * (KC / 16) + kk);  // the chunk's last tap group
        }
    }
    }  // !RING
    if constexpr (BLK) {  // (a chain the table left open counts as a block; then the total takes the accumulators' place)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = tot[i][j] + acc[i][j];
    }

    if constexpr (G2 > 0) {
        // ---- fused trailing 1x1 (ConvArgs::w2) ---------------------------------------------------------------------
        // WM == 1 and TM == cout_pad: this wave's accumulators hold every channel of t for its 16*NT pixels.  In the
        // MFMA result layout lane group q holds channels 4q..4q+3 of a 16-channel tile in the four accumulator
        // registers -- exactly what the B operand of k-step e (register e) of the next GEMM needs, and exactly the
        // channel order (e, 4+e, 8+e, 12+e; e = 0..3) the stand-alone kernel's ds_read_b128 fragments produce.  So t
        // never leaves the registers, and each y element is the same fma chain as in the unfused pair of launches.
        // The second layer's weights come through LDS in groups of G2 cout tiles (double-buffered, register-staged);
        // each group's outputs leave through the same LDS transpose as the ordinary epilogue.
        static_assert(WM == 1 && KC == 16, "fused tail: one wave row, 16-channel stages");
        constexpr int SLAB4 = MT * G2 * 64;  // 16-byte slots of one weight group: [MT chunks][16*G2 rows][4]
        constexpr int WU = (SLAB4 + 255) / 256;
        constexpr int SW2 = 16 * G2 + 4;  // staging row stride (floats)
        constexpr int S42 = 4 * G2;       // float4 per staged pixel row
        constexpr int EU2 = TP * S42 / 256;
        static_assert(TP * S42 % 256 == 0, "fused epilogue tiling");
        float* w2l = smem;                  // [2][SLAB4 * 4]
        float* est = smem + 2 * SLAB4 * 4;  // [TP][SW2]
        const int K2 = a.cout_pad;          // reduction length of the second layer (= TM)
        const int ng2 = a.cout2_pad / (16 * G2);
        // LEAD: a third GEMM u = relu(w3 * y + bias3) (the next block's leading 1x1, TM couts again) rides along: each
        // finished group of y (16*G2 channels, final values, read back from the staging tile in B-fragment layout) is one
        // slice of its reduction, so u accumulates group by group in the order the stand-alone launch walks its chunks.
        constexpr int SLAB3 = G2 * MT * 64;  // 16-byte slots of one w3 slice: [G2 chunks][TM rows][4]
        constexpr int WU3 = (SLAB3 + 255) / 256;
        float* w3l = est + TP * SW2;         // [2][SLAB3 * 4]
        const int K3 = a.cout2_pad;
        unsigned w3_off[LEAD ? WU3 : 1];
        f32x4 pw3[LEAD ? WU3 : 1];
        f32x4 uacc[LEAD ? MT : 1][LEAD ? NT : 1];
        if constexpr (LEAD) {
#pragma unroll
            for (int u = 0; u < WU3; ++u) {
                const int f = tid + u * 256;
                const int c = f / (TM * 4), row = (f >> 2) % TM, c4 = f & 3;
                w3_off[u] = (unsigned)(row * K3 + c * 16 + c4 * 4) * 4u;
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int k = 0; k < NT; ++k)
                    uacc[i][k] = a.tail_bias_init ? *reinterpret_cast<const f32x4*>(gbias3 + i * 16 + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        auto issue_w3 = [&](int g) {
            if constexpr (LEAD) {
                const char* base = reinterpret_cast<const char*>(gw3) + (size_t)g * (G2 * 16) * 4;  // column slice g
#pragma unroll
                for (int u = 0; u < WU3; ++u)
                    if (tid + u * 256 < SLAB3) pw3[u] = *reinterpret_cast<const f32x4*>(base + w3_off[u]);
            }
        };
        auto commit_w3 = [&](int buf) {
            if constexpr (LEAD) {
#pragma unroll
                for (int u = 0; u < WU3; ++u)
                    if (tid + u * 256 < SLAB3)
                        *reinterpret_cast<f32x4*>(w3l + buf * (SLAB3 * 4) + (tid + u * 256) * 4) = pw3[u];
            }
        };
        unsigned w2_off[WU];
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int f = tid + u * 256;
            const int c = f / (G2 * 64), row = (f >> 2) % (G2 * 16), c4 = f & 3;
            w2_off[u] = (unsigned)(row * K2 + c * 16 + c4 * 4) * 4u;
        }
        f32x4 pw2[WU];
        auto issue_w2 = [&](int g) {
            const char* base = reinterpret_cast<const char*>(gw2) + (size_t)g * (G2 * 16) * K2 * 4;  // wave-uniform
#pragma unroll
            for (int u = 0; u < WU; ++u)
                if (tid + u * 256 < SLAB4) pw2[u] = *reinterpret_cast<const f32x4*>(base + w2_off[u]);
        };
        auto commit_w2 = [&](int buf) {
#pragma unroll
            for (int u = 0; u < WU; ++u)
                if (tid + u * 256 < SLAB4)
                    *reinterpret_cast<f32x4*>(w2l + buf * (SLAB4 * 4) + (tid + u * 256) * 4) = pw2[u];
        };
        issue_w2(0);
        issue_w3(0);
        // t = act_mid(acc + bias): the float operations of the stand-alone launch's epilogue
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(gbias + i * 16 + q * 4);
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                f32x4 t = a.bias_mode == 0 ? acc[i][k] + b4 : acc[i][k];
                if (a.act_mid == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = fmaxf(t[e], 0.f);
                }
                acc[i][k] = t;
            }
        }
        __syncthreads();  // every wave has left the main loop: the stage buffers are free
        commit_w2(0);
        commit_w3(0);
        __syncthreads();
        const size_t img_px2 = (size_t)n * a.OH * a.OW;
        char* yn2 = reinterpret_cast<char*>(gy + img_px2 * a.ycs);
        const char* r1n2 = reinterpret_cast<const char*>(gres1 + img_px2 * a.r1cs);
        for (int g = 0; g < ng2; ++g) {
            if (g + 1 < ng2) {
                issue_w2(g + 1);
                issue_w3(g + 1);
            }
            // the residual operand of this group's outputs is requested before the group's MFMAs (it depends on the pixel
            // and the channel group only): its round trip hides behind them instead of sitting in front of the stores
            constexpr bool PRE = EU2 <= 4;
            f32x4 r1pre[PRE ? EU2 : 1];
            // ... and so is the second layer's bias of this group (a global load that used to sit between the staging read and
            // the add of every write-back batch): one float4 per thread when all its slots share a channel quad (256 % S42 == 0)
            constexpr bool BPRE = PRE && 256 % S42 == 0;
            f32x4 b2pre = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (BPRE) b2pre = *reinterpret_cast<const f32x4*>(gbias2 + g * (G2 * 16) + (tid % S42) * 4);
            if constexpr (PRE) {
#pragma unroll
                for (int u = 0; u < EU2; ++u) {
                    const int f = tid + u * 256;
                    const int p = f / S42, c4 = f - p * S42;
                    const int cb = g * (G2 * 16) + c4 * 4;
                    const int gy = ty0 + (p >> tw_log2), gx = tx0 + (p & (TW - 1));
                    const bool okp = cb < a.cout_store && gy < a.GH && gx < a.GW;
                    r1pre[u] = (okp && a.res1)
                                   ? *reinterpret_cast<const f32x4*>(r1n2 + (size_t)(((unsigned)(gy * a.OW + gx) * (unsigned)a.r1cs + cb) * 4u))
                                   : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            f32x4 acc2[G2][NT];
#pragma unroll
            for (int i = 0; i < G2; ++i)
#pragma unroll
                for (int k = 0; k < NT; ++k)
                    acc2[i][k] = a.tail_bias_init ? *reinterpret_cast<const f32x4*>(gbias2 + g * (G2 * 16) + i * 16 + q * 4)
                                                  : (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* cw = w2l + (g & 1) * (SLAB4 * 4);
#pragma unroll
            for (int c = 0; c < MT; ++c) {
                f32x4 af2[G2];
#pragma unroll
                for (int i = 0; i < G2; ++i)
                    af2[i] = *reinterpret_cast<const f32x4*>(cw + ((c * G2 + i) * 16 + l15) * 16 + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < G2; ++i)
#pragma unroll
                        for (int k = 0; k < NT; ++k)
                            acc2[i][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(af2[i][e], acc[c][k][e], acc2[i][k], 0, 0, 0);
            }
            if (g > 0) __syncthreads();  // everyone has read the previous group out of the staging tile
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int p = (wn * NT + k) * 16 + l15;
#pragma unroll
                for (int i = 0; i < G2; ++i) *reinterpret_cast<f32x4*>(est + p * SW2 + i * 16 + q * 4) = acc2[i][k];
            }
            if (g + 1 < ng2) {  // those buffers were last read two barriers ago
                commit_w2((g + 1) & 1);
                commit_w3((g + 1) & 1);
            }
            __syncthreads();
            constexpr int UB2 = EU2 % 4 == 0 ? 4 : (EU2 % 3 == 0 ? 3 : (EU2 % 2 == 0 ? 2 : 1));
            for (int u0 = 0; u0 < EU2; u0 += UB2) {
                f32x4 r1[UB2], v[UB2];
                unsigned pixs[UB2];
                int cbs[UB2];
                bool ok[UB2];
#pragma unroll
                for (int u = 0; u < UB2; ++u) {
                    const int f = tid + (u0 + u) * 256;
                    const int p = f / S42, c4 = f - p * S42;
                    const int cb = g * (G2 * 16) + c4 * 4;
                    const int gy = ty0 + (p >> tw_log2), gx = tx0 + (p & (TW - 1));
                    ok[u] = cb < a.cout_store && gy < a.GH && gx < a.GW;
                    cbs[u] = cb;
                    pixs[u] = (unsigned)(gy * a.OW + gx);
                    v[u] = *reinterpret_cast<const f32x4*>(est + p * SW2 + c4 * 4);
                    if constexpr (PRE) {
                        r1[u] = r1pre[u0 + u];
                    } else {
                        r1[u] = (ok[u] && a.res1)
                                    ? *reinterpret_cast<const f32x4*>(r1n2 + (size_t)((pixs[u] * (unsigned)a.r1cs + cb) * 4u))
                                    : (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int u = 0; u < UB2; ++u) {
                    if (!ok[u]) continue;
                    f32x4 w;
                    if (a.tail_bias_init) w = v[u];  // (the chain started from the bias)
                    else if constexpr (BPRE) w = v[u] + b2pre;
                    else w = v[u] + *reinterpret_cast<const f32x4*>(gbias2 + cbs[u]);
                    if (a.res1) w += r1[u];
                    if (a.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = fmaxf(w[e], 0.f);
                    } else if (a.act == ACT_LEAKY) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = w[e] > 0.f ? w[e] : w[e] * 0.01f;
                    }
                    *reinterpret_cast<f32x4*>(yn2 + (size_t)((pixs[u] * (unsigned)a.ycs + cbs[u]) * 4u)) = w;
                    if constexpr (LEAD) v[u] = w;
                }
                if constexpr (LEAD) {  // the final y values go back into the slots they came from (same thread)
#pragma unroll
                    for (int u = 0; u < UB2; ++u) {
                        const int f = tid + (u0 + u) * 256;
                        const int p = f / S42, c4 = f - p * S42;
                        *reinterpret_cast<f32x4*>(est + p * SW2 + c4 * 4) = v[u];
                    }
                }
            }
            if constexpr (LEAD) {
                __syncthreads();
                const float* cw3 = w3l + (g & 1) * (SLAB3 * 4);
#pragma unroll
                for (int c = 0; c < G2; ++c) {
                    f32x4 af3[MT], bf3[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) af3[i] = *reinterpret_cast<const f32x4*>(cw3 + ((c * MT + i) * 16 + l15) * 16 + q * 4);
#pragma unroll
                    for (int k = 0; k < NT; ++k)
                        bf3[k] = *reinterpret_cast<const f32x4*>(est + ((wn * NT + k) * 16 + l15) * SW2 + c * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int k = 0; k < NT; ++k)
                                uacc[i][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(af3[i][e], bf3[k][e], uacc[i][k], 0, 0, 0);
                }
            }
        }
        if constexpr (LEAD) {
            // u = relu(uacc + bias3) leaves through an LDS transpose like any other output tile
            constexpr int SW3 = 16 * MT + 4, S43 = 4 * MT, EU3 = TP * S43 / 256;
            static_assert(TP * S43 % 256 == 0, "lead epilogue tiling");
            __syncthreads();  // the last group's fragments have been read
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(gbias3 + i * 16 + q * 4);
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    f32x4 t = a.tail_bias_init ? uacc[i][k] : uacc[i][k] + b4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = fmaxf(t[e], 0.f);
                    *reinterpret_cast<f32x4*>(smem + ((wn * NT + k) * 16 + l15) * SW3 + i * 16 + q * 4) = t;
                }
            }
            __syncthreads();
            char* y3n = reinterpret_cast<char*>(gy3 + img_px2 * a.y3cs);
#pragma unroll
            for (int u = 0; u < EU3; ++u) {
                const int f = tid + u * 256;
                const int p = f / S43, c4 = f - p * S43;
                const int gy = ty0 + (p >> tw_log2), gx = tx0 + (p & (TW - 1));
                if (gy < a.GH && gx < a.GW)
                    *reinterpret_cast<f32x4*>(y3n + (size_t)(((unsigned)(gy * a.OW + gx) * (unsigned)a.y3cs + c4 * 4) * 4u)) =
                        *reinterpret_cast<const f32x4*>(smem + p * SW3 + c4 * 4);
            }
        }
        return;
    }

    // Epilogue.  The MFMA result layout gives each lane 4 consecutive couts of one pixel (64-byte segments per pixel
    // and store instruction); writing that straight to HBM wastes half of every 128-byte line transaction.  The tile is
    // therefore staged through LDS as [pixel][cout] and written back by all 256 threads with consecutive lanes on
    // consecutive couts of the same pixel (full lines for TM >= 32 couts), which is also how the fused
    // residual / gate / skip operands are read.
    const size_t img_px = (size_t)n * a.OH * a.OW;  // wave-uniform per-image bases (byte pointers)
    char* yn = reinterpret_cast<char*>(gy + img_px * a.ycs);
    const char* r1n = reinterpret_cast<const char*>(gres1 + img_px * a.r1cs);
    const char* mln = reinterpret_cast<const char*>(gmul + img_px * a.mcs);
    const char* r2n = reinterpret_cast<const char*>(gres2 + img_px * a.r2cs);
    char* y2n = reinterpret_cast<char*>(gy2 + img_px * a.y2cs);
    char* ptn = reinterpret_cast<char*>(gpartial + ((size_t)split * a.N * a.OH * a.OW + img_px) * a.cout_pad);
    for (int ip = 0; ip < MT; ip += EMT) {
        __syncthreads();  // LDS is free: the last stage (or the previous pass) has been consumed
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int p = (wn * NT + k) * 16 + l15;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (i >= ip && i < ip + EMT)
                    *reinterpret_cast<f32x4*>(smem + p * SW + (wm * EMT + (i - ip)) * 16 + q * 4) = acc[i][k];
        }
        __syncthreads();
        // EU elements (float4) per thread, handled UB at a time: the fused operands of a batch are all requested before
        // the first one is used, so the epilogue costs EU/UB memory round trips instead of EU
        constexpr int UB = EU % 6 == 0 ? 6 : (EU % 4 == 0 ? 4 : (EU % 3 == 0 ? 3 : (EU % 2 == 0 ? 2 : 1)));
        static_assert(TP * S4 % 256 == 0, "epilogue tiling");
        for (int u0 = 0; u0 < EU; u0 += UB) {
            f32x4 r1[UB], ml[UB], r2[UB], v[UB];
            unsigned pixs[UB];  // pixel index inside image n: every operand is addressed as a wave-uniform per-image
            int cbs[UB];        // base + a 32-bit byte offset (no 64-bit VALU multiplies per slot)
            bool ok[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int f = tid + (u0 + u) * 256;
                const int p = f / S4, c4 = f - p * S4;
                const int wmc = c4 / (4 * EMT), ii = (c4 / 4) % EMT;
                const int cb = co0 + (wmc * MT + ip + ii) * 16 + (c4 & 3) * 4;
                const int gy = ty0 + (p >> tw_log2);
                const int gx = tx0 + (ck ? 2 * (p & (TW - 1)) + ((gy & 1) ^ (a.ckbd == 1 ? 1 : 0)) : (p & (TW - 1)));
                ok[u] = ip + ii < MT && cb < a.cout_store && gy < a.GH && gx < a.GW;
                cbs[u] = cb;
                // sub-pixel form of a stride-2 transposed conv with <= 4 couts (ConvArgs::subpix): the 16 computed channels
                // are 4 output phases x 4 channels, so this float4 belongs to output pixel (2gy + py, 2gx + px), channels 0..3
                const int oy = a.subpix ? ((cb >> 3) & 1) : oy_off, ox = a.subpix ? ((cb >> 2) & 1) : ox_off;
                pixs[u] = (unsigned)((gy * a.OS + oy) * a.OW + (gx * a.OS + ox));
                v[u] = *reinterpret_cast<const f32x4*>(smem + p * SW + c4 * 4);
                const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                r1[u] = (ok[u] && a.res1) ? *reinterpret_cast<const f32x4*>(r1n + (size_t)((pixs[u] * (unsigned)a.r1cs + cb) * 4u)) : z;
                ml[u] = (ok[u] && a.mul) ? *reinterpret_cast<const f32x4*>(mln + (size_t)((pixs[u] * (unsigned)a.mcs + cb) * 4u)) : z;
                r2[u] = (ok[u] && a.res2) ? *reinterpret_cast<const f32x4*>(r2n + (size_t)((pixs[u] * (unsigned)a.r2cs + cb) * 4u)) : z;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (!ok[u]) continue;
                const int cb = cbs[u];
                f32x4 w = v[u];
                if (a.partial) {  // raw partial sums; bias / activation / fused operands are applied by the reducer
                    *reinterpret_cast<f32x4*>(ptn + (size_t)((pixs[u] * (unsigned)a.cout_pad + cb) * 4u)) = w;
                    continue;
                }
                if (a.bias_mode == 0) w += *reinterpret_cast<const f32x4*>(gbias + cb);
                if (a.res1) w += r1[u];
                if (a.act == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = fmaxf(w[e], 0.f);
                } else if (a.act == ACT_LEAKY) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = w[e] > 0.f ? w[e] : w[e] * 0.01f;
                } else if (a.act == ACT_SIGMOID) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = a.exact_math ? rgbd_sigmoid_ref(w[e]) : sigmoid_f32(w[e]);
                }  // ACT_GELU never reaches this kernel's epilogue: launch_conv routes it through the reducer (the erff
                   // expansion, unrolled for every output slot, used to be 2/3 of this kernel's code)
                if (a.mul) w *= ml[u];
                if (a.res2) w += r2[u];
                *reinterpret_cast<f32x4*>(yn + (size_t)((pixs[u] * (unsigned)a.ycs + (a.subpix ? 0 : cb)) * 4u)) = w;
                if (a.y2) *reinterpret_cast<f32x4*>(y2n + (size_t)((pixs[u] * (unsigned)a.y2cs + cb) * 4u)) = w;
            }
        }
    }
}

template <int WM, int WN, int MT, int NT, int KC, bool DMA, int NB = 2, bool BLK = false>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a, int tw_log2, int tiles_x, int tiles_y,
                                                         int taps_per_stage, int tab_f)
{
    conv_mfma_body<WM, WN, MT, NT, KC, DMA, 0, false, NB, BLK>(a, tw_log2, tiles_x, tiles_y, taps_per_stage, tab_f);
}

// The fused variants are compiled for two workgroups per CU (256 registers per lane): left alone, the allocator parks the
// first layer's accumulators in AGPRs and copies them into VGPRs for the second GEMM (302 registers, one workgroup per CU).
template <int WM, int WN, int MT, int NT, int KC, bool DMA, int G2, bool LEAD, bool BLK = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_mfma_kernel_fused(
    ConvArgs a, int tw_log2, int tiles_x, int tiles_y, int taps_per_stage, int tab_f)
{
    conv_mfma_body<WM, WN, MT, NT, KC, DMA, G2, LEAD, 2, BLK>(a, tw_log2, tiles_x, tiles_y, taps_per_stage, tab_f);
}

namespace {

constexpr int LDS_BUDGET = 78 * 1024;  // two workgroups per CU (160 KiB LDS)

template <int WM, int WN, int MT, int NT, int KC, bool DMA, int G2 = 0, bool LEAD = false, int NB = 2, bool BLK = false>
int launch_cfg(const ConvArgs& a, int tw_log2, hipStream_t s, int lds_cap = 0)
{
    const long budget = lds_cap > 0 ? lds_cap : LDS_BUDGET;
    constexpr int TM = 16 * MT * WM;
    constexpr int TP = 16 * NT * WN;
    constexpr int RS = KC > 16 ? KC + 4 : KC;
    const int TW = 1 << tw_log2, TH = TP / TW;
    const int TWx = a.ckbd ? 2 * TW : TW;  // columns a tile spans (checkerboard output: every second one is computed)
    const int tiles_x = (a.GW + TWx - 1) / TWx, tiles_y = (a.GH + TH - 1) / TH;
    const int PH = (TH - 1) * a.IS + a.span_y, PW = (TWx - 1) * a.IS + a.span_x;
    const size_t patch_bytes = (size_t)PH * PW * RS * sizeof(float);
    const size_t tap_bytes = (size_t)TM * RS * sizeof(float);
    int max_taps = 1;
    for (int p = 0; p < a.nphase; ++p) max_taps = a.taps.n[p] > max_taps ? a.taps.n[p] : max_taps;
    constexpr int PR = TP >= 128 ? 12 : (TP >= 64 ? 6 : 4);
    if ((long)PH * PW * (KC / 4) > PR * 256) return RGBD_ENOSPC;  // patch registers
    // ring staging (NB > 2): single-tap stride-1 layers without halo, whole waves of patch slots
    if (NB > 2 && (max_taps != 1 || a.nphase != 1 || a.IS != 1 || a.ckbd || a.span_y != 1 || a.span_x != 1 || TP % 64)) return RGBD_ENOSPC;
    long room = (8 * 256) / ((long)TM * (KC / 4));               // weight slots per stage (WR)
    if (DMA && NB == 2) {
        const long lds_room = (budget - 2 * (long)patch_bytes) / (2 * (long)tap_bytes);
        room = room < lds_room ? room : lds_room;
    }
    if (room < 1) return RGBD_ENOSPC;
    const int tps = (int)(room < max_taps ? room : max_taps);
    if (KC > 16 && tps < max_taps) return RGBD_ENOSPC;  // would break the canonical accumulation order
    constexpr int EMT = (TP * (16 * WM * MT + 4) * 4 <= 64 * 1024) ? MT : (MT + 1) / 2;
    const size_t epi_bytes = (size_t)TP * (16 * WM * EMT + 4) * sizeof(float);
    const size_t stage_bytes = (patch_bytes + (size_t)tps * tap_bytes) * (DMA ? NB : 1);
    // fused tail: two weight-group slabs + the output staging tile (it replaces the ordinary epilogue)
    size_t fuse_bytes = G2 > 0 ? (size_t)2 * MT * G2 * 1024 + (size_t)TP * (16 * G2 + 4) * sizeof(float) : 0;
    if (LEAD) {  // + two slices of the third layer's weights; the u tile is staged over everything at the end
        fuse_bytes += (size_t)2 * G2 * MT * 1024;
        const size_t u_bytes = (size_t)TP * (16 * MT + 4) * sizeof(float);
        fuse_bytes = fuse_bytes > u_bytes ? fuse_bytes : u_bytes;
        if (!a.w3 || !a.bias3 || !a.y3 || a.cout3_pad != TM || a.y3cs % 4) return RGBD_EINVAL;
    }
    const size_t tail_bytes = G2 > 0 ? fuse_bytes : epi_bytes;
    const size_t buf_bytes = ((stage_bytes > tail_bytes ? stage_bytes : tail_bytes) + 15) & ~(size_t)15;
    const size_t lds = buf_bytes + 256;  // + the workgroup's expanded tap table (2 x 32 ints)
    if (G2 > 0 && (lds > (size_t)LDS_BUDGET + 256 || a.cout_pad != TM || a.cout2_pad % (16 * (G2 > 0 ? G2 : 1)))) return RGBD_ENOSPC;
    void (*kern)(ConvArgs, int, int, int, int, int);
    if constexpr (G2 > 0) kern = conv_mfma_kernel_fused<WM, WN, MT, NT, KC, DMA, G2, LEAD, BLK>;
    else kern = conv_mfma_kernel<WM, WN, MT, NT, KC, DMA, NB, BLK>;
    if (lds > 64 * 1024) {  // the attribute is per device: one flag per (instantiation, device), set under a lock
        static std::mutex mu;
        static bool configured[64] = {false};
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !configured[dev]) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(160 * 1024)));
            if (dev >= 0 && dev < 64) configured[dev] = true;
        }
    }
    dim3 grid((unsigned)(tiles_x * tiles_y * a.N * (a.groups == 2 ? 2 : 1)) * (unsigned)((a.cout_pad + TM - 1) / TM), 1,
              (unsigned)(a.nphase * a.splitk));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, tw_log2, tiles_x, tiles_y, tps, (int)(buf_bytes / 4));
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

}  // namespace

struct Choice {
    int wm, mt, nt, kc, tw_log2;
    bool dma;
    int lds_cap = 0;  // 0: LDS_BUDGET (two workgroups per CU); else a smaller cap (52 KiB: three per CU, 38 KiB: four)
    int ring = 0;     // 4 / 3: ring of that many DMA stage buffers (single-tap layers; staging mode 4 / 5 of the tile table)
};

// single-tap stride-1 layer without halo: eligible for ring staging
inline bool ring_ok(const ConvArgs& a)
{
    int max_taps = 1;
    for (int p = 0; p < a.nphase; ++p) max_taps = a.taps.n[p] > max_taps ? a.taps.n[p] : max_taps;
    return max_taps == 1 && a.nphase == 1 && a.IS == 1 && !a.ckbd && a.span_y == 1 && a.span_x == 1;
}

inline void set_mode(Choice& c, int kc, int dm)
{
    c.kc = kc;
    c.dma = dm != 0 && kc == 16;
    c.lds_cap = dm == 2 ? 52 * 1024 : (dm == 3 ? 38 * 1024 : 0);
    c.ring = (kc == 16 && dm == 4) ? 4 : ((kc == 16 && dm == 5) ? 3 : 0);
}
