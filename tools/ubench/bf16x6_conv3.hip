// Third stage of the split-bf16 experiment: a whole 3x3 stride-1 layer with the design the second stage pointed at
// -- 32-channel stages, the fp32 patch split into three bf16 planes ONCE at staging time (every staged element is used by all
// nine taps), weights split once beforehand, six v_mfma_f32_16x16x32_bf16 per fp32-equivalent product block.
//   X [N][H][W][C] fp32, Wt [Co][9][C] fp32 (tap = 3 dy + dx, zero padding 1), Y [N][H][W][Co] = relu(conv + bias)
//   workgroup: 8 x 32 output pixels x 48 output channels, 4 waves (wave w: rows 2w, 2w+1 = four 16-pixel tiles x three
//   16-channel tiles); LDS: patch 10 x 34 pixels x 3 planes x 64 B + weights 9 taps x 48 x 3 planes x 64 B = 145 KB.
//   16-byte chunks (8 channels) of a pixel / weight row are XOR-swizzled by ((index >> 2) & 3): conflict-free ds_read_b128.
//   hipcc --offload-arch=gfx950 -O2 bf16x6_conv3.hip -o bf16x6_conv3 && ./bf16x6_conv3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __host__ inline unsigned f2u(float x) { unsigned u; memcpy(&u, &x, 4); return u; }
__device__ __host__ inline float u2f(unsigned u) { float x; memcpy(&x, &u, 4); return x; }
__device__ __forceinline__ void split3(float v, u16& h, u16& m, u16& l)
{
    unsigned u = f2u(v);
    u += 0x7FFFu + ((u >> 16) & 1u);
    u &= 0xFFFF0000u;
    h = (u16)(u >> 16);
    v -= u2f(u);
    u = f2u(v);
    u += 0x7FFFu + ((u >> 16) & 1u);
    u &= 0xFFFF0000u;
    m = (u16)(u >> 16);
    v -= u2f(u);
    u = f2u(v);
    u += 0x7FFFu + ((u >> 16) & 1u);
    l = (u16)(u >> 16);
}
__global__ void split_kernel(const float* __restrict__ x, u16* __restrict__ planes, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u16 h, m, l;
        split3(x[i], h, m, l);
        planes[i] = h;
        planes[n + i] = m;
        planes[2 * n + i] = l;
    }
}

// weights as the kernel's LDS image, so that staging them is a linear copy by LDS DMA:
// img[coblk][chunk][plane][row = tap * COT + co][64 B: 8-channel pieces at (kb ^ ((co >> 2) & 3)) * 16]
__global__ void weight_image_kernel(const float* __restrict__ w, u16* __restrict__ img, int Co, int C)
{
    const int nchunk = C / 32, ncb = Co / 48;
    const size_t total = (size_t)ncb * nchunk * 9 * 48 * 32;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 32), co = (int)((i / 32) % 48), tap = (int)((i / (32 * 48)) % 9);
        const int chunk = (int)((i / (32 * 48 * 9)) % nchunk), cb = (int)(i / ((size_t)32 * 48 * 9 * nchunk));
        u16 p[3];
        split3(w[((size_t)(cb * 48 + co) * 9 + tap) * C + chunk * 32 + ch], p[0], p[1], p[2]);
        const int kb = ch >> 3;
        const size_t row = (size_t)tap * 48 + co;
        for (int pl = 0; pl < 3; ++pl)
            img[((((size_t)cb * nchunk + chunk) * 3 + pl) * (9 * 48) + row) * 32 + ((kb ^ ((co >> 2) & 3)) * 8) + (ch & 7)] = p[pl];
    }
}

#define TH 8
#define TW 32
#define PW (TW + 2)
#define NPX ((TH + 2) * PW)  // 340 patch pixels
#define COT 48
#define PATCH_B (NPX * 64)       // one plane
#define WTS_B (9 * COT * 64)     // one plane
__global__ __launch_bounds__(256) void conv3_bf16x6(const float* __restrict__ X, const u16* __restrict__ Wp, const float* __restrict__ bias,
                                                    float* __restrict__ Y, int N, int H, int W, int C, int Co)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* patch = lds;                 // [3][NPX][64]
    unsigned char* wts = lds + 3 * PATCH_B;     // [3][9][COT][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int t = blockIdx.x, n = t / (tiles_x * tiles_y), ty = (t / tiles_x) % tiles_y, tx = t % tiles_x;
    const int y0 = ty * TH, x0 = tx * TW, co0 = blockIdx.y * COT;
    const size_t nW = (size_t)Co * 9 * C;
    f32x4 acc[3][4];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // patch staging, fixed per thread for all chunks: item u = (patch pixel, 4-channel piece); global offset (or -1 outside the
    // image) and LDS offset computed once
    constexpr int PU = (NPX * 8 + 255) / 256;  // 11
    int goff[PU], loff[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int f = tid + u * 256;
        const int c4 = f & 7, ppx = f >> 3, py = ppx / PW, px = ppx % PW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool in = f < NPX * 8 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[u] = in ? (int)((((size_t)n * H + gy) * W + gx) * C + c4 * 4) : -1;
        loff[u] = f < NPX * 8 ? ppx * 64 + (((c4 >> 1) ^ ((ppx >> 2) & 3)) * 16) + (c4 & 1) * 8 : -1;
    }
    f32x4 pre[PU];
    auto gload = [&](int c0) {
#pragma unroll
        for (int u = 0; u < PU; ++u) pre[u] = goff[u] >= 0 ? *reinterpret_cast<const f32x4*>(X + goff[u] + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    const int nchunk = C / 32;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid) >> 6;
    gload(0);
    for (int c0 = 0, chunk = 0; c0 < C; c0 += 32, ++chunk) {
        __syncthreads();  // the previous chunk's fragments have been read
        // weights of this chunk: a linear 3 x 27,648-byte copy by LDS DMA (1 KB per wave-load)
        {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(Wp) + ((size_t)blockIdx.y * nchunk + chunk) * (3 * WTS_B);
            for (int w = wave_u; w < 3 * WTS_B / 1024; w += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + w * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(wts + w * 1024), 16, 0, 0);
        }
        // patch of this chunk from the prefetched registers: split into the three planes here, once
#pragma unroll
        for (int u = 0; u < PU; ++u)
            if (loff[u] >= 0) {
                u16x4 h, m, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u16 a, b, c;
                    split3(pre[u][e], a, b, c);
                    h[e] = a;
                    m[e] = b;
                    l[e] = c;
                }
                *reinterpret_cast<u16x4*>(patch + 0 * PATCH_B + loff[u]) = h;
                *reinterpret_cast<u16x4*>(patch + 1 * PATCH_B + loff[u]) = m;
                *reinterpret_cast<u16x4*>(patch + 2 * PATCH_B + loff[u]) = l;
            }
        if (c0 + 32 < C) gload(c0 + 32);  // the next chunk's patch travels under this chunk's MFMAs
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PU) : "memory");  // the weight DMA (older than the prefetch) has landed
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            bf16x8 a[3][3], b[4][3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int co = i * 16 + l15;
                const int off = (tap * COT + co) * 64 + ((q ^ ((co >> 2) & 3)) * 16);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[i][pl] = *reinterpret_cast<const bf16x8*>(wts + pl * WTS_B + off);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ppx = (2 * wave + (j >> 1) + dy) * PW + (j & 1) * 16 + l15 + dx;
                const int off = ppx * 64 + ((q ^ ((ppx >> 2) & 3)) * 16);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[j][pl] = *reinterpret_cast<const bf16x8*>(patch + pl * PATCH_B + off);
            }
            // the six products term by term over all twelve accumulators (smallest first): a product never waits for the one
            // before it on the same accumulator
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA[term]], b[j][PB[term]], acc[i][j], 0, 0, 0);
            }
        }
    }
    // D: column = pixel l15 of tile j, rows 4q .. 4q+3 = output channels of tile i
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co0 + i * 16 + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gy = y0 + 2 * wave + (j >> 1), gx = x0 + (j & 1) * 16 + l15;
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[i][j][r] + bv[r], 0.f);
            *reinterpret_cast<f32x4*>(Y + (((size_t)n * H + gy) * W + gx) * Co + co0 + i * 16 + 4 * q) = o;
        }
    }
}


// ---- second cut: 16-channel stages so that two workgroups share a CU (74 KB of LDS each) and hide each other's staging; one
// MFMA step (K = 32) covers TWO taps x 16 channels: lane groups q = 0, 1 supply tap A's channels, q = 2, 3 tap B's (the ninth
// tap is paired with zeros).  Weight image: [coblk][chunk16][plane][tap * 48 + co][32 B: pieces at (kb ^ ((co >> 3) & 1)) * 16].
__global__ void weight_image16_kernel(const float* __restrict__ w, u16* __restrict__ img, int Co, int C)
{
    const int nchunk = C / 16, ncb = Co / 48;
    const size_t total = (size_t)ncb * nchunk * 9 * 48 * 16;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 16), co = (int)((i / 16) % 48), tap = (int)((i / (16 * 48)) % 9);
        const int chunk = (int)((i / (16 * 48 * 9)) % nchunk), cb = (int)(i / ((size_t)16 * 48 * 9 * nchunk));
        u16 p[3];
        split3(w[((size_t)(cb * 48 + co) * 9 + tap) * C + chunk * 16 + ch], p[0], p[1], p[2]);
        const int kb = ch >> 3;
        const size_t row = (size_t)tap * 48 + co;
        for (int pl = 0; pl < 3; ++pl)
            img[((((size_t)cb * nchunk + chunk) * 3 + pl) * (9 * 48) + row) * 16 + ((kb ^ ((co >> 3) & 1)) * 8) + (ch & 7)] = p[pl];
    }
}
#define PATCH16_B (NPX * 32)
#define WTS16_B (9 * COT * 32)
__global__ __launch_bounds__(256, 2) void conv3_bf16x6_v2(const float* __restrict__ X, const u16* __restrict__ Wp, const float* __restrict__ bias,
                                                          float* __restrict__ Y, int N, int H, int W, int C, int Co)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* patch = lds;                   // [3][NPX][32]
    unsigned char* wts = lds + 3 * PATCH16_B;     // [3][9][COT][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4, kb = q & 1, hiq = q >> 1;
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int t = blockIdx.x, n = t / (tiles_x * tiles_y), ty = (t / tiles_x) % tiles_y, tx = t % tiles_x;
    const int y0 = ty * TH, x0 = tx * TW, co0 = blockIdx.y * COT;
    f32x4 acc[3][4];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int PU = (NPX * 4 + 255) / 256;  // 6
    int goff[PU], loff[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int f = tid + u * 256;
        const int c4 = f & 3, ppx = f >> 2, py = ppx / PW, px = ppx % PW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool in = f < NPX * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[u] = in ? (int)((((size_t)n * H + gy) * W + gx) * C + c4 * 4) : -1;
        loff[u] = f < NPX * 4 ? ppx * 32 + (((c4 >> 1) ^ ((ppx >> 3) & 1)) * 16) + (c4 & 1) * 8 : -1;
    }
    f32x4 pre[PU];
    auto gload = [&](int c0) {
#pragma unroll
        for (int u = 0; u < PU; ++u) pre[u] = goff[u] >= 0 ? *reinterpret_cast<const f32x4*>(X + goff[u] + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    const int nchunk = C / 16;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid) >> 6;
    // fragment addresses that do not depend on the chunk
    int aoff[3], boff[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int co = i * 16 + l15;
        aoff[i] = co * 32 + ((kb ^ ((co >> 3) & 1)) * 16);  // + tap * COT * 32
    }
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    gload(0);
    for (int c0 = 0, chunk = 0; c0 < C; c0 += 16, ++chunk) {
        __syncthreads();
        {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(Wp) + ((size_t)blockIdx.y * nchunk + chunk) * (3 * WTS16_B);
            for (int w = wave_u; w * 1024 < 3 * WTS16_B; w += 4)
                if (w * 1024 + lane * 16 < 3 * WTS16_B)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + w * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(wts + w * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < PU; ++u)
            if (loff[u] >= 0) {
                u16x4 h, m, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u16 a, b, c;
                    split3(pre[u][e], a, b, c);
                    h[e] = a;
                    m[e] = b;
                    l[e] = c;
                }
                *reinterpret_cast<u16x4*>(patch + 0 * PATCH16_B + loff[u]) = h;
                *reinterpret_cast<u16x4*>(patch + 1 * PATCH16_B + loff[u]) = m;
                *reinterpret_cast<u16x4*>(patch + 2 * PATCH16_B + loff[u]) = l;
            }
        if (c0 + 16 < C) gload(c0 + 16);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PU) : "memory");
        __syncthreads();
#pragma unroll 1
        for (int st = 0; st < 5; ++st) {
            const int tap = 2 * st + hiq;           // this lane group's tap (9 = none)
            const int tp = tap < 9 ? tap : 8;        // (address only)
            const int dy = tp / 3, dx = tp % 3;
            bf16x8 a[3][3], b[4][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[i][pl] = *reinterpret_cast<const bf16x8*>(wts + pl * WTS16_B + tp * COT * 32 + aoff[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ppx = (2 * wave + (j >> 1) + dy) * PW + (j & 1) * 16 + l15 + dx;
                const int off = ppx * 32 + ((kb ^ ((ppx >> 3) & 1)) * 16);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + pl * PATCH16_B + off);
                    b[j][pl] = tap < 9 ? v : zero8;
                }
            }
            // the six products term by term over all twelve accumulators (smallest first): a product never waits for the one
            // before it on the same accumulator
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA[term]], b[j][PB[term]], acc[i][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co0 + i * 16 + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gy = y0 + 2 * wave + (j >> 1), gx = x0 + (j & 1) * 16 + l15;
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[i][j][r] + bv[r], 0.f);
            *reinterpret_cast<f32x4*>(Y + (((size_t)n * H + gy) * W + gx) * Co + co0 + i * 16 + 4 * q) = o;
        }
    }
}


// ---- third cut: what the counters of the first two asked for (MFMA busy 37 / 41 %, 44 % of the LDS cycles bank conflicts,
// 5,100 vector instructions per wave, fragment reads and MFMAs strictly one after the other):
//   * LDS layout [plane][8-channel block kb][pixel or (tap, channel) row][16 B] with every kb region a multiple of 256 B: a
//     ds_read_b128 lane group is 8 lanes of one kb and 8 of the next, all with different rows -- 16 consecutive rows, one bank
//     quad each, whatever the kb;
//   * the tap loop fully unrolled: a tap is an immediate offset on per-(tile, plane) base addresses computed once;
//   * the next tap's fragments are read while the current tap's 72 MFMAs run (two register sets).
// One workgroup per CU (8 x 32 pixels x 48 channels, 32-channel stages as in the first cut).
#define PR3 5632   // bytes per (plane, kb) region of the patch: 340 pixels x 16 B, rounded up to 256
#define WR3 6912   // bytes per (plane, kb) region of the weights: 9 taps x 48 channels x 16 B (= 27 x 256)
__global__ void weight_image3_kernel(const float* __restrict__ w, u16* __restrict__ img, int Co, int C)
{
    const int nchunk = C / 32, ncb = Co / 48;
    const size_t total = (size_t)ncb * nchunk * 9 * 48 * 32;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 32), co = (int)((i / 32) % 48), tap = (int)((i / (32 * 48)) % 9);
        const int chunk = (int)((i / (32 * 48 * 9)) % nchunk), cb = (int)(i / ((size_t)32 * 48 * 9 * nchunk));
        u16 p[3];
        split3(w[((size_t)(cb * 48 + co) * 9 + tap) * C + chunk * 32 + ch], p[0], p[1], p[2]);
        const int kb = ch >> 3;
        for (int pl = 0; pl < 3; ++pl)  // image of one (coblk, chunk): [plane][kb][tap * 48 + co][8 bf16]
            img[(((size_t)cb * nchunk + chunk) * 12 + pl * 4 + kb) * (WR3 / 2) + (size_t)(tap * 48 + co) * 8 + (ch & 7)] = p[pl];
    }
}
__global__ __launch_bounds__(256) void conv3_bf16x6_v3(const float* __restrict__ X, const u16* __restrict__ Wp, const float* __restrict__ bias,
                                                       float* __restrict__ Y, int N, int H, int W, int C, int Co)
{
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    unsigned char* patch = lds;              // [3][4][PR3]
    unsigned char* wts = lds + 12 * PR3;     // [3][4][WR3]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int t = blockIdx.x, n = t / (tiles_x * tiles_y), ty = (t / tiles_x) % tiles_y, tx = t % tiles_x;
    const int y0 = ty * TH, x0 = tx * TW, co0 = blockIdx.y * COT;
    f32x4 acc[3][4];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // patch staging: item u = (patch pixel, 8-channel block): two float4 loads, one 16-byte store per plane
    constexpr int PU = (NPX * 4 + 255) / 256;  // 6
    int goff[PU], loff[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int f = tid + u * 256;
        const int kb = f & 3, ppx = f >> 2, py = ppx / PW, px = ppx % PW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool in = f < NPX * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[u] = in ? (int)((((size_t)n * H + gy) * W + gx) * C + kb * 8) : -1;
        loff[u] = f < NPX * 4 ? kb * PR3 + ppx * 16 : -1;
    }
    f32x4 pre[PU][2];
    auto gload = [&](int c0) {
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            pre[u][0] = goff[u] >= 0 ? *reinterpret_cast<const f32x4*>(X + goff[u] + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
            pre[u][1] = goff[u] >= 0 ? *reinterpret_cast<const f32x4*>(X + goff[u] + c0 + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    const int nchunk = C / 32;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid) >> 6;
    // fragment base addresses (plane 0; a plane is + 4 * region): weights [kb = q][co = 16 i + l15], patch [kb = q][pixel]
    const unsigned char* abase = wts + q * WR3 + l15 * 16;
    const unsigned char* bbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bbase[j] = patch + q * PR3 + ((2 * wave + (j >> 1)) * PW + (j & 1) * 16 + l15) * 16;
    gload(0);
    for (int c0 = 0, chunk = 0; c0 < C; c0 += 32, ++chunk) {
        __syncthreads();
        {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(Wp) + ((size_t)blockIdx.y * nchunk + chunk) * (12 * WR3);
            for (int w = wave_u; w < 12 * WR3 / 1024; w += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + w * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(wts + w * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < PU; ++u)
            if (loff[u] >= 0) {
                unsigned hw[4], mw[4], lw[4];  // 8 bf16 per plane as 4 dwords
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    u16 a0, b0, c0_, a1, b1, c1;
                    split3(pre[u][e >> 2][e & 3], a0, b0, c0_);
                    split3(pre[u][e >> 2][(e & 3) + 1], a1, b1, c1);
                    hw[e >> 1] = (unsigned)a0 | ((unsigned)a1 << 16);
                    mw[e >> 1] = (unsigned)b0 | ((unsigned)b1 << 16);
                    lw[e >> 1] = (unsigned)c0_ | ((unsigned)c1 << 16);
                }
                *reinterpret_cast<uint4*>(patch + 0 * 4 * PR3 + loff[u]) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                *reinterpret_cast<uint4*>(patch + 1 * 4 * PR3 + loff[u]) = make_uint4(mw[0], mw[1], mw[2], mw[3]);
                *reinterpret_cast<uint4*>(patch + 2 * 4 * PR3 + loff[u]) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
            }
        if (c0 + 32 < C) gload(c0 + 32);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PU) : "memory");  // the weight DMA (older than the prefetch) has landed
        __syncthreads();
        bf16x8 a[2][3][3], b[2][4][3];
        auto frag = [&](int set, int tap) {
            const int toff = ((tap / 3) * PW + (tap % 3)) * 16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < 3; ++i) a[set][i][pl] = *reinterpret_cast<const bf16x8*>(abase + pl * 4 * WR3 + (tap * COT + i * 16) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[set][j][pl] = *reinterpret_cast<const bf16x8*>(bbase[j] + pl * 4 * PR3 + toff);
            }
        };
        frag(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1;
            if (tap + 1 < 9) frag(cur ^ 1, tap + 1);
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[cur][i][PA[term]], b[cur][j][PB[term]], acc[i][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co0 + i * 16 + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gy = y0 + 2 * wave + (j >> 1), gx = x0 + (j & 1) * 16 + l15;
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[i][j][r] + bv[r], 0.f);
            *reinterpret_cast<f32x4*>(Y + (((size_t)n * H + gy) * W + gx) * Co + co0 + i * 16 + 4 * q) = o;
        }
    }
}


// ---- fourth cut: the third cut's layout and unrolling with 16-channel stages and tap pairs (second cut), so that TWO workgroups
// of 80 KB share a CU and overlap each other's start-up, staging and epilogue.  K = 32 of one MFMA step = tap A x 16 channels
// (lane groups q = 0, 1) + tap B x 16 channels (q = 2, 3); the ninth tap pairs with a tenth, all-zero weight tap.
#define WR4 7680   // bytes per (plane, kb) region of the weights: 10 taps x 48 channels x 16 B (= 30 x 256)
__global__ void weight_image4_kernel(const float* __restrict__ w, u16* __restrict__ img, int Co, int C)
{
    const int nchunk = C / 16, ncb = Co / 48;
    const size_t total = (size_t)ncb * nchunk * 10 * 48 * 16;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 16), co = (int)((i / 16) % 48), tap = (int)((i / (16 * 48)) % 10);
        const int chunk = (int)((i / (16 * 48 * 10)) % nchunk), cb = (int)(i / ((size_t)16 * 48 * 10 * nchunk));
        u16 p[3] = {0, 0, 0};
        if (tap < 9) split3(w[((size_t)(cb * 48 + co) * 9 + tap) * C + chunk * 16 + ch], p[0], p[1], p[2]);
        const int kb = ch >> 3;
        for (int pl = 0; pl < 3; ++pl)  // image of one (coblk, chunk): [plane][kb][tap * 48 + co][8 bf16]
            img[(((size_t)cb * nchunk + chunk) * 6 + pl * 2 + kb) * (WR4 / 2) + (size_t)(tap * 48 + co) * 8 + (ch & 7)] = p[pl];
    }
}
template <int KO>  // knock-outs for timing only (results wrong): 1 no split arithmetic, 2 no patch staging after the first chunk,
                    // 3 no weight DMA after the first chunk, 4 fragments read once per chunk only, 5 = 2 + 3 + 4
__global__ __launch_bounds__(256, 2) void conv3_bf16x6_v4(const float* __restrict__ X, const u16* __restrict__ Wp, const float* __restrict__ bias,
                                                          float* __restrict__ Y, int N, int H, int W, int C, int Co)
{
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    unsigned char* patch = lds;             // [3][2][PR3]
    unsigned char* wts = lds + 6 * PR3;     // [3][2][WR4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4, kb = q & 1, hiq = q >> 1;
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int t = blockIdx.x, n = t / (tiles_x * tiles_y), ty = (t / tiles_x) % tiles_y, tx = t % tiles_x;
    const int y0 = ty * TH, x0 = tx * TW, co0 = blockIdx.y * COT;
    f32x4 acc[3][4];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int PU = (NPX * 2 + 255) / 256;  // 3: item = (patch pixel, 8-channel block)
    int goff[PU], loff[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int f = tid + u * 256;
        const int kbs = f & 1, ppx = f >> 1, py = ppx / PW, px = ppx % PW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool in = f < NPX * 2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[u] = in ? (int)((((size_t)n * H + gy) * W + gx) * C + kbs * 8) : -1;
        loff[u] = f < NPX * 2 ? kbs * PR3 + ppx * 16 : -1;
    }
    f32x4 pre[PU][2];
    auto gload = [&](int c0) {
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            pre[u][0] = goff[u] >= 0 ? *reinterpret_cast<const f32x4*>(X + goff[u] + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
            pre[u][1] = goff[u] >= 0 ? *reinterpret_cast<const f32x4*>(X + goff[u] + c0 + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    const int nchunk = C / 16;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid) >> 6;
    // per-lane fragment bases per step: this lane group's tap of the step (2 st + hiq; tap 9 = the zero tap, patch offset 0)
    const unsigned char *ab[5], *bb[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int tap = 2 * st + hiq, tp = tap < 9 ? tap : 0;
        ab[st] = wts + kb * WR4 + (tap * COT + l15) * 16;
        bb[st] = patch + kb * PR3 + ((2 * wave + tp / 3) * PW + (tp % 3) + l15) * 16;
    }
    if (KO >= 10) {  // de-phasing experiment: every other workgroup (by bit KO - 10 of its index) starts ~3 us late
        if ((blockIdx.x >> (KO - 10)) & 1)
            __builtin_amdgcn_s_sleep(100);  // 6,400 cycles
    }
    gload(0);
    for (int c0 = 0, chunk = 0; c0 < C; c0 += 16, ++chunk) {
        __syncthreads();
        {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(Wp) + ((size_t)blockIdx.y * nchunk + chunk) * (6 * WR4);
            if (!((KO == 3 || KO == 5) && chunk > 0))
            for (int w = wave_u; w < 6 * WR4 / 1024; w += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + w * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(wts + w * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < PU; ++u)
            if (loff[u] >= 0 && !((KO == 2 || KO == 5) && chunk > 0)) {
                unsigned hw[4], mw[4], lw[4];
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    u16 a0, b0, c0_, a1, b1, c1;
                    if (KO == 1) {
                        a0 = b0 = c0_ = (u16)(f2u(pre[u][e >> 2][e & 3]) >> 16);
                        a1 = b1 = c1 = (u16)(f2u(pre[u][e >> 2][(e & 3) + 1]) >> 16);
                    } else {
                        split3(pre[u][e >> 2][e & 3], a0, b0, c0_);
                        split3(pre[u][e >> 2][(e & 3) + 1], a1, b1, c1);
                    }
                    hw[e >> 1] = (unsigned)a0 | ((unsigned)a1 << 16);
                    mw[e >> 1] = (unsigned)b0 | ((unsigned)b1 << 16);
                    lw[e >> 1] = (unsigned)c0_ | ((unsigned)c1 << 16);
                }
                *reinterpret_cast<uint4*>(patch + 0 * 2 * PR3 + loff[u]) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                *reinterpret_cast<uint4*>(patch + 1 * 2 * PR3 + loff[u]) = make_uint4(mw[0], mw[1], mw[2], mw[3]);
                *reinterpret_cast<uint4*>(patch + 2 * 2 * PR3 + loff[u]) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
            }
        if (c0 + 16 < C && !(KO == 2 || KO == 5)) gload(c0 + 16);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PU) : "memory");
        __syncthreads();
        bf16x8 a[3][3], b[4][3];
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            if (!((KO == 4 || KO == 5) && st > 0))
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < 3; ++i) a[i][pl] = *reinterpret_cast<const bf16x8*>(ab[st] + pl * 2 * WR4 + i * 16 * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j][pl] = *reinterpret_cast<const bf16x8*>(bb[st] + pl * 2 * PR3 + ((j >> 1) * PW + (j & 1) * 16) * 16);
            }
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA[term]], b[j][PB[term]], acc[i][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co0 + i * 16 + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gy = y0 + 2 * wave + (j >> 1), gx = x0 + (j & 1) * 16 + l15;
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[i][j][r] + bv[r], 0.f);
            *reinterpret_cast<f32x4*>(Y + (((size_t)n * H + gy) * W + gx) * Co + co0 + i * 16 + 4 * q) = o;
        }
    }
}

int main()
{
    // (H a multiple of 8, W of 32, C of 32, Co of 48: the experiment has no edge tiles)
    struct { const char* name; int N, H, W, C, Co; } shapes[] = {{"4 x 128x160, 3x3 96 -> 96", 4, 128, 160, 96, 96}, {"4 x 256x320, 3x3 96 -> 96", 4, 256, 320, 96, 96},
                                                                 {"4 x 64x96, 3x3 96 -> 96", 4, 64, 96, 96, 96},     {"4 x 32x64, 3x3 160 -> 144", 4, 32, 64, 160, 144}};
    const size_t lds = 3 * PATCH_B + 3 * WTS_B;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_bf16x6), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_bf16x6_v2), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * PATCH16_B + 3 * WTS16_B);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_bf16x6_v3), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * PR3 + 12 * WR3);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_bf16x6_v4<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * PR3 + 6 * WR4);
    srand(3);
    for (auto& s : shapes) {
        const size_t nX = (size_t)s.N * s.H * s.W * s.C, nW = (size_t)s.Co * 9 * s.C, nY = (size_t)s.N * s.H * s.W * s.Co;
        float *hX = (float*)malloc(nX * 4), *hW = (float*)malloc(nW * 4), *hb = (float*)malloc(s.Co * 4), *hY = (float*)malloc(nY * 4);
        for (size_t i = 0; i < nX; ++i) hX[i] = (float)((rand() % 2001) - 1000) / 500.f * (1.f + (rand() % 97) * 1e-4f);
        for (size_t i = 0; i < nW; ++i) hW[i] = (float)((rand() % 2001) - 1000) / 20000.f * (1.f + (rand() % 89) * 1e-4f);
        for (int i = 0; i < s.Co; ++i) hb[i] = 0.01f * i;
        float *X, *Wt, *b, *Y;
        u16* Wp;
        (void)hipMalloc(&X, nX * 4); (void)hipMalloc(&Wt, nW * 4); (void)hipMalloc(&b, s.Co * 4); (void)hipMalloc(&Y, nY * 4); (void)hipMalloc(&Wp, nW * 7);
        (void)hipMemcpy(X, hX, nX * 4, hipMemcpyHostToDevice); (void)hipMemcpy(Wt, hW, nW * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(b, hb, s.Co * 4, hipMemcpyHostToDevice);
        const dim3 grid(s.N * (s.H / TH) * (s.W / TW), s.Co / COT);
        const size_t lds2 = 3 * PATCH16_B + 3 * WTS16_B;
        const size_t lds3 = 12 * PR3 + 12 * WR3;
        const size_t lds4 = 6 * PR3 + 6 * WR4;
        float ms_v[4];
        double err_v[4];
        for (int ver = 0; ver < 4; ++ver) {
            if (ver == 0) hipLaunchKernelGGL(weight_image_kernel, dim3(512), dim3(256), 0, 0, Wt, Wp, s.Co, s.C);
            else if (ver == 1) hipLaunchKernelGGL(weight_image16_kernel, dim3(512), dim3(256), 0, 0, Wt, Wp, s.Co, s.C);
            else if (ver == 2) hipLaunchKernelGGL(weight_image3_kernel, dim3(512), dim3(256), 0, 0, Wt, Wp, s.Co, s.C);
            else hipLaunchKernelGGL(weight_image4_kernel, dim3(512), dim3(256), 0, 0, Wt, Wp, s.Co, s.C);
            auto launch = [&]() {
                if (ver == 0) hipLaunchKernelGGL(conv3_bf16x6, grid, dim3(256), lds, 0, X, Wp, b, Y, s.N, s.H, s.W, s.C, s.Co);
                else if (ver == 1) hipLaunchKernelGGL(conv3_bf16x6_v2, grid, dim3(256), lds2, 0, X, Wp, b, Y, s.N, s.H, s.W, s.C, s.Co);
                else if (ver == 2) hipLaunchKernelGGL(conv3_bf16x6_v3, grid, dim3(256), lds3, 0, X, Wp, b, Y, s.N, s.H, s.W, s.C, s.Co);
                else hipLaunchKernelGGL(conv3_bf16x6_v4<0>, grid, dim3(256), lds4, 0, X, Wp, b, Y, s.N, s.H, s.W, s.C, s.Co);
            };
            (void)hipMemset(Y, 0, nY * 4);
            launch();
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            for (int r = 0; r < 10; ++r) launch();
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            ms_v[ver] = ms / 10;
            (void)hipMemcpy(hY, Y, nY * 4, hipMemcpyDeviceToHost);
            double err = 0;
            srand(11);
            for (int tsm = 0; tsm < 3000; ++tsm) {
                const int n = rand() % s.N, y = tsm < 200 ? (tsm & 1 ? 0 : s.H - 1) : rand() % s.H, x = tsm < 200 ? (tsm & 2 ? 0 : s.W - 1) : rand() % s.W, co = rand() % s.Co;
                double ref = hb[co], mag = 0;
                for (int dy = 0; dy < 3; ++dy)
                    for (int dx = 0; dx < 3; ++dx) {
                        const int gy = y + dy - 1, gx = x + dx - 1;
                        if (gy < 0 || gy >= s.H || gx < 0 || gx >= s.W) continue;
                        for (int c = 0; c < s.C; ++c) {
                            const double p = (double)hX[(((size_t)n * s.H + gy) * s.W + gx) * s.C + c] * hW[((size_t)co * 9 + dy * 3 + dx) * s.C + c];
                            ref += p;
                            mag += fabs(p);
                        }
                    }
                ref = ref > 0 ? ref : 0;
                err = fmax(err, fabs(hY[(((size_t)n * s.H + y) * s.W + x) * s.Co + co] - ref) / mag);
            }
            err_v[ver] = err;
        }
        const double gf = 2.0 * s.N * s.H * s.W * s.Co * 9.0 * s.C / 1e9;
        printf("%-28s first cut: %7.1f us %6.1f TFLOP/s (err %.1e) | second (16-ch stages, tap pairs, 2 WG/CU): %7.1f us %6.1f (err %.1e) | third (conflict-free layout, unrolled taps, fragment prefetch): %7.1f us %6.1f (err %.1e) | fourth (third's layout, 16-ch stages, 2 WG/CU): %7.1f us %6.1f (err %.1e)\n",
               s.name, ms_v[0] * 1e3, gf / ms_v[0], err_v[0], ms_v[1] * 1e3, gf / ms_v[1], err_v[1], ms_v[2] * 1e3, gf / ms_v[2], err_v[2], ms_v[3] * 1e3, gf / ms_v[3], err_v[3]);
        if (s.H == 256) {  // knock-outs of the fourth cut on the large layer: what each part of the loop costs
            auto tko = [&](auto kern, const char* what) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * PR3 + 6 * WR4);
                hipLaunchKernelGGL(kern, grid, dim3(256), lds4, 0, X, Wp, b, Y, s.N, s.H, s.W, s.C, s.Co);
                (void)hipDeviceSynchronize();
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0);
                for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, grid, dim3(256), lds4, 0, X, Wp, b, Y, s.N, s.H, s.W, s.C, s.Co);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                printf("    fourth cut, %-62s %7.1f us\n", what, ms * 100);
            };
            tko(conv3_bf16x6_v4<0>, "as is");
            tko(conv3_bf16x6_v4<1>, "without the split arithmetic");
            tko(conv3_bf16x6_v4<2>, "patch staged for the first chunk only");
            tko(conv3_bf16x6_v4<3>, "weights staged for the first chunk only");
            tko(conv3_bf16x6_v4<4>, "fragments read for the first step of a chunk only");
            tko(conv3_bf16x6_v4<5>, "all three: MFMAs, barriers, prologue and epilogue");
            tko(conv3_bf16x6_v4<10>, "as is, workgroups with bit 0 of their index set start 2.7 us late");
            tko(conv3_bf16x6_v4<11>, "... bit 1");
            tko(conv3_bf16x6_v4<13>, "... bit 3");
            tko(conv3_bf16x6_v4<14>, "... bit 4");
            tko(conv3_bf16x6_v4<18>, "... bit 8");
            tko(conv3_bf16x6_v4<19>, "... bit 9");
        }
        (void)hipFree(X); (void)hipFree(Wt); (void)hipFree(b); (void)hipFree(Y); (void)hipFree(Wp);
        free(hX); free(hW); free(hb); free(hY);
    }
    return 0;
}
