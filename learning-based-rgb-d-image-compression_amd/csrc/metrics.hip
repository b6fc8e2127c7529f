// MS-SSIM statistics on the GPU for the harness (reference utils/metrics.py:8-14 calls pytorch_msssim.ms_ssim; the
// definition that package implements -- 11-tap Gaussian (sigma 1.5) "valid" filtering of x, y, x^2, y^2, xy per channel, the
// SSIM and contrast-structure maps, their means, five dyadic scales with 2x2 average pooling in between -- is restated in
// rgbd_amd/metrics.py on torch ops and, in fp64, in oracle/msssim_ref.py).
//
// Why a kernel: the torch restatement is ~60 small launches per image and modality, enqueued under the GIL; with eight or
// sixteen harness workers in flight that enqueue was most of the job's wall time (profiles/r04_harness_throughput.txt).  Here
// one call per batch of planes does all five scales: per scale one kernel (tile staged in LDS, horizontal then vertical
// 11-tap pass over the five maps, per-workgroup partial sums in a fixed order) plus one pooling kernel, and a final
// reducer that adds the partials in workgroup order -- deterministic, no atomics.
#include "common.h"

namespace {

constexpr int KT = 11;           // taps
constexpr int TH = 16, TW = 64;  // output tile of one workgroup
constexpr int IH = TH + KT - 1, IW = TW + KT - 1;

struct Taps {
    float g[KT];
};

// x, y: [P][H][W] planes (P = N * C); out rows r < H - 10, cols c < W - 10.  partial[(p * nblk + blk) * 2 + {0, 1}] = sums of
// the SSIM / CS maps over this workgroup's tile.
__global__ __launch_bounds__(256) void msssim_scale_kernel(const float* __restrict__ x, const float* __restrict__ y, int H, int W,
                                                           int clamp01, Taps tp, float c1, float c2, int tiles_x,
                                                           float* __restrict__ partial)
{
    __shared__ float xs[IH][IW + 1], ys[IH][IW + 1];
    __shared__ float hx[5][IH][TW + 1];
    __shared__ float red[2][256];
    const int tid = threadIdx.x;
    const int p = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const int OH = H - (KT - 1), OW = W - (KT - 1);
    const float* xp = x + (size_t)p * H * W;
    const float* yp = y + (size_t)p * H * W;
    for (int i = tid; i < IH * IW; i += 256) {
        const int r = i / IW, c = i - r * IW;
        const int gy = ty0 + r, gx = tx0 + c;
        float a = 0.f, b = 0.f;
        if (gy < H && gx < W) {
            a = xp[(size_t)gy * W + gx];
            b = yp[(size_t)gy * W + gx];
            if (clamp01) {
                a = fminf(fmaxf(a, 0.f), 1.f);
                b = fminf(fmaxf(b, 0.f), 1.f);
            }
        }
        xs[r][c] = a;
        ys[r][c] = b;
    }
    __syncthreads();
    for (int i = tid; i < IH * TW; i += 256) {  // horizontal pass: five maps
        const int r = i / TW, c = i - r * TW;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const float a = xs[r][c + k], b = ys[r][c + k], g = tp.g[k];
            s0 = fmaf(g, a, s0);
            s1 = fmaf(g, b, s1);
            s2 = fmaf(g, a * a, s2);
            s3 = fmaf(g, b * b, s3);
            s4 = fmaf(g, a * b, s4);
        }
        hx[0][r][c] = s0;
        hx[1][r][c] = s1;
        hx[2][r][c] = s2;
        hx[3][r][c] = s3;
        hx[4][r][c] = s4;
    }
    __syncthreads();
    float sum_s = 0.f, sum_c = 0.f;
    for (int i = tid; i < TH * TW; i += 256) {  // vertical pass + the maps
        const int r = i / TW, c = i - r * TW;
        if (ty0 + r >= OH || tx0 + c >= OW) continue;
        float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const float g = tp.g[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] = fmaf(g, hx[q][r + k][c], m[q]);
        }
        const float mu1 = m[0], mu2 = m[1];
        const float s1 = m[2] - mu1 * mu1, s2 = m[3] - mu2 * mu2, s12 = m[4] - mu1 * mu2;
        const float cs = (2.f * s12 + c2) / (s1 + s2 + c2);
        const float ss = ((2.f * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs;
        sum_s += ss;
        sum_c += cs;
    }
    red[0][tid] = sum_s;
    red[1][tid] = sum_c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {  // fixed tree
        if (tid < o) {
            red[0][tid] += red[0][tid + o];
            red[1][tid] += red[1][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const size_t k = ((size_t)p * gridDim.x + blockIdx.x) * 2;
        partial[k] = red[0][0];
        partial[k + 1] = red[1][0];
    }
}

// out[(p * 5 + scale) * 2 + {0, 1}] = mean SSIM / mean CS of plane p at this scale: the partials added in workgroup order
__global__ void msssim_finish_kernel(const float* __restrict__ partial, int nblk, float inv_count, int scale, int P,
                                     float* __restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float s = 0.f, c = 0.f;
    for (int b = 0; b < nblk; ++b) {
        s += partial[((size_t)p * nblk + b) * 2];
        c += partial[((size_t)p * nblk + b) * 2 + 1];
    }
    out[((size_t)p * 5 + scale) * 2] = s * inv_count;
    out[((size_t)p * 5 + scale) * 2 + 1] = c * inv_count;
}

// F.avg_pool2d(x, 2, padding = (H % 2, W % 2)) (count_include_pad: the divisor is always 4); optional clamp of the inputs
__global__ void avgpool2_kernel(const float* __restrict__ x, int H, int W, int ph, int pw, int OH, int OW, int clamp01,
                                float* __restrict__ y, size_t total)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const size_t p = i / ((size_t)OW * OH);
        const float* xp = x + p * H * W;
        float s = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int iy = 2 * oy - ph + dy, ix = 2 * ox - pw + dx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    float v = xp[(size_t)iy * W + ix];
                    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
                    s += v;
                }
            }
        y[i] = s * 0.25f;
    }
}

}  // namespace

// Bytes of device workspace rgbd_msssim_stats needs for P planes of H x W.
extern "C" int64_t rgbd_msssim_workspace_bytes(int32_t P, int32_t H, int32_t W)
{
    if (P <= 0 || H <= (KT - 1) * 16 || W <= (KT - 1) * 16) return -1;
    size_t floats = 0;
    int h = H, w = W;
    for (int sc = 0; sc < 5; ++sc) {  // the walk of rgbd_msssim_stats: partial sums of the scale, then the next scale's two images
        floats += (size_t)P * ((h - (KT - 1) + TH - 1) / TH) * ((w - (KT - 1) + TW - 1) / TW) * 2;
        if (sc < 4) {
            h = (h + 2 * (h % 2) - 2) / 2 + 1;
            w = (w + 2 * (w % 2) - 2) / 2 + 1;
            floats += (size_t)2 * P * h * w;
        }
    }
    return (int64_t)(floats * sizeof(float) + 256);
}

// x, y: device [P][H][W] fp32 (contiguous planes: P = N * C).  out: device [P][5][2] = per plane and scale the mean of the
// SSIM map and of the contrast-structure map (what pytorch_msssim's _ssim returns per channel); the caller combines them
// (relu, weights, product over scales, mean over channels -- metrics.py).  clamp01: clamp both inputs to [0, 1] first
// (utils/metrics.py:9-10).  min(H, W) must exceed 160 (five scales of an 11-tap filter).
extern "C" int rgbd_msssim_stats(const float* x, const float* y, int32_t P, int32_t H, int32_t W, const float* taps11,
                                 float data_range, int32_t clamp01, float* out, void* workspace, int64_t workspace_bytes,
                                 void* stream)
{
    if (!x || !y || !taps11 || !out || !workspace || P <= 0 || P > 65535) return RGBD_EINVAL;
    if ((H < W ? H : W) <= (KT - 1) * 16) return RGBD_EINVAL;
    if (workspace_bytes < rgbd_msssim_workspace_bytes(P, H, W)) return RGBD_ENOSPC;
    hipStream_t s = (hipStream_t)stream;
    Taps tp;
    for (int k = 0; k < KT; ++k) tp.g[k] = taps11[k];
    const float c1 = (0.01f * data_range) * (0.01f * data_range), c2 = (0.03f * data_range) * (0.03f * data_range);
    float* ws = (float*)workspace;
    const float *cx = x, *cy = y;
    int h = H, w = W;
    int cl = clamp01 ? 1 : 0;
    for (int sc = 0; sc < 5; ++sc) {
        const int tiles_x = (w - (KT - 1) + TW - 1) / TW, tiles_y = (h - (KT - 1) + TH - 1) / TH;
        const int nblk = tiles_x * tiles_y;
        float* partial = ws;
        float* next = partial + (size_t)P * nblk * 2;
        hipLaunchKernelGGL(msssim_scale_kernel, dim3(nblk, P), dim3(256), 0, s, cx, cy, h, w, cl, tp, c1, c2, tiles_x, partial);
        hipLaunchKernelGGL(msssim_finish_kernel, dim3((P + 63) / 64), dim3(64), 0, s, partial, nblk,
                           1.0f / ((float)(h - (KT - 1)) * (float)(w - (KT - 1))), sc, P, out);
        if (sc < 4) {
            const int ph = h % 2, pw = w % 2;
            const int oh = (h + 2 * ph - 2) / 2 + 1, ow = (w + 2 * pw - 2) / 2 + 1;
            float* nx = next;
            float* ny = nx + (size_t)P * oh * ow;
            const size_t total = (size_t)P * oh * ow;
            size_t g = (total + 255) / 256;
            if (g > 4096) g = 4096;
            hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)g), dim3(256), 0, s, cx, h, w, ph, pw, oh, ow, cl, nx, total);
            hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)g), dim3(256), 0, s, cy, h, w, ph, pw, oh, ow, cl, ny, total);
            cx = nx;
            cy = ny;
            h = oh;
            w = ow;
            cl = 0;  // (already clamped)
            ws = ny + (size_t)P * oh * ow;
        }
    }
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}
