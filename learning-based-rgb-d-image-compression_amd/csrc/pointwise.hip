// HBM-bound helper kernels of the ELIC_united path (gfx950): layout conversion at the API boundary, the ESA pooled
// branch (max_pool2d 7/3 + bilinear upsample, modules/transform/attention.py:87-92), SE_Block (global average pool
// -> 2 bias-free FCs -> sigmoid, attention.py:63-67) and channel scaling / concatenation.
// All tensors are NHWC with channel stride cs (multiple of 4 floats): every access is a 16-byte vector per lane and
// consecutive lanes walk consecutive channels, so each wave instruction touches contiguous 1 KiB runs.
#include "common.h"
#include "exact_math.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline unsigned grid_for(size_t work, int block = 256, unsigned cap = 256 * 8)
{
    size_t g = (work + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc16_kernel(const float* __restrict__ src, int N, int C, int H, int W, float* __restrict__ dst,
                                      int cs, int perm)
{
    const size_t npix = (size_t)N * H * W;
    const int c4n = cs / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix * c4n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / c4n;
        const int c4 = (int)(i - pix * c4n);
        const size_t hw = pix % ((size_t)H * W);
        const size_t n = pix / ((size_t)H * W);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = rgbd_cperm(c4 * 4 + e, perm);  // the logical channel stored at this physical position
            v[e] = c < C ? src[(n * C + c) * (size_t)H * W + hw] : 0.f;
        }
        *reinterpret_cast<f32x4*>(dst + pix * cs + c4 * 4) = v;
    }
}

int launch_nchw_to_nhwc16(const float* src, int N, int C, int H, int W, float* dst, int cs, hipStream_t s, int perm)
{
    const size_t work = (size_t)N * H * W * (cs / 4);
    hipLaunchKernelGGL(nchw_to_nhwc16_kernel, dim3(grid_for(work)), dim3(256), 0, s, src, N, C, H, W, dst, cs, perm);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, int N, int C, int H, int W, int cs,
                                    float* __restrict__ dst, int clamp01, int perm)
{
    const size_t total = (size_t)N * C * H * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t hw = i % ((size_t)H * W);
        const size_t nc = i / ((size_t)H * W);
        const int c = (int)(nc % C);
        const size_t n = nc / C;
        float v = src[(n * H * W + hw) * cs + rgbd_cperm(c, perm)];
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        dst[i] = v;
    }
}

int launch_nhwc_to_nchw_clamp(const float* src, int N, int C, int H, int W, int cs, float* dst, int clamp01,
                              hipStream_t s, int perm)
{
    const size_t work = (size_t)N * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(work)), dim3(256), 0, s, src, N, C, H, W, cs, dst, clamp01, perm);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// ---------------------------------------------------------------------------------------------
// F.max_pool2d(kernel=7, stride=3), no padding, floor mode (attention.py:87)
__global__ void maxpool7s3_kernel(const float* __restrict__ x0, int N, int H, int W, int cs, float* __restrict__ y0,
                                  int OH, int OW, const float* __restrict__ x1, float* __restrict__ y1)
{
    // blockIdx.y = 1: the second tensor of a pair (the other modality's ESA branch, same shape) in the same launch
    const float* __restrict__ x = blockIdx.y ? x1 : x0;
    float* __restrict__ y = blockIdx.y ? y1 : y0;
    const int c4n = cs / 4;
    const size_t total = (size_t)N * OH * OW * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        size_t t = i / c4n;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const size_t n = t / OH;
        f32x4 m = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int ky = 0; ky < 7; ++ky)
            for (int kx = 0; kx < 7; ++kx) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(
                    x + ((n * H + (oy * 3 + ky)) * (size_t)W + (ox * 3 + kx)) * cs + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        *reinterpret_cast<f32x4*>(y + ((n * OH + oy) * (size_t)OW + ox) * cs + c4 * 4) = m;
    }
}

int launch_maxpool7s3(const float* x, int N, int H, int W, int cs, float* y, int OH, int OW, hipStream_t s, const float* x1,
                      float* y1)
{
    if (OH != (H - 7) / 3 + 1 || OW != (W - 7) / 3 + 1 || H < 7 || W < 7 || !x1 != !y1) return RGBD_EINVAL;
    const size_t work = (size_t)N * OH * OW * (cs / 4);
    hipLaunchKernelGGL(maxpool7s3_kernel, dim3(grid_for(work), x1 ? 2 : 1), dim3(256), 0, s, x, N, H, W, cs, y, OH, OW, x1, y1);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// F.interpolate(mode="bilinear", align_corners=False) to (H, W) (attention.py:91):
// src = max(0, scale*(dst+0.5)-0.5), i0 = floor(src), i1 = min(i0+1, in-1), l1 = src - i0, l0 = 1 - l1
// ref != 0: the arithmetic of torch's CPU kernels (UpSampleKernel.cpp, third-party to the reference; restated from black-box
// probes, oracle/cpu_arith.c orc_bilinear): the source index is ONE fused multiply-add, and the four taps are combined
//   * output H + W > 128 (the generic separable kernel): t_k = fma(v_k0, lx0, v_k1 * lx1);  out = fma(t_0, ly0, t_1 * ly1)
//   * output H + W <= 128 (the channels-last vector kernel): w_ij = ly_i * lx_j, and
//       channels below C - C % 16 (its vector body):  fma(w00, v00, fma(w01, v01, fma(w11, v11, w10 * v10)))
//       the C % 16 tail channels (its scalar loop):    fma(w11, v11, fma(w10, v10, fma(w00, v00, w01 * v01)))
// ref: 1 + C (the tensor's channel count; channels are stored permuted, rgbd_cperm)
__global__ void bilinear_kernel(const float* __restrict__ x0, int N, int h, int w, int cs, float* __restrict__ y0, int H,
                                int W, float sy, float sx, const float* __restrict__ x1, float* __restrict__ y1, int ref)
{
    const float* __restrict__ x = blockIdx.y ? x1 : x0;  // (pair of tensors, see maxpool7s3_kernel)
    float* __restrict__ y = blockIdx.y ? y1 : y0;
    const int c4n = cs / 4;
    const size_t total = (size_t)N * H * W * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        size_t t = i / c4n;
        const int ox = (int)(t % W);
        t /= W;
        const int oy = (int)(t % H);
        const size_t n = t / H;
        float fy = ref ? __fmaf_rn(sy, (float)oy + 0.5f, -0.5f) : __fsub_rn(__fmul_rn(sy, (float)oy + 0.5f), 0.5f);
        float fx = ref ? __fmaf_rn(sx, (float)ox + 0.5f, -0.5f) : __fsub_rn(__fmul_rn(sx, (float)ox + 0.5f), 0.5f);
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* b = x + n * (size_t)h * w * cs + c4 * 4;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * w + x0) * cs);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * w + x1) * cs);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * w + x0) * cs);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * w + x1) * cs);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!ref) {
                const float top = __fadd_rn(__fmul_rn(lx0, v00[e]), __fmul_rn(lx1, v01[e]));
                const float bot = __fadd_rn(__fmul_rn(lx0, v10[e]), __fmul_rn(lx1, v11[e]));
                o[e] = __fadd_rn(__fmul_rn(ly0, top), __fmul_rn(ly1, bot));
            } else if (H + W > 128) {
                const float top = __fmaf_rn(v00[e], lx0, __fmul_rn(v01[e], lx1));
                const float bot = __fmaf_rn(v10[e], lx0, __fmul_rn(v11[e], lx1));
                o[e] = __fmaf_rn(top, ly0, __fmul_rn(bot, ly1));
            } else {
                const float w00 = __fmul_rn(ly0, lx0), w01 = __fmul_rn(ly0, lx1), w10 = __fmul_rn(ly1, lx0), w11 = __fmul_rn(ly1, lx1);
                const int C = ref - 1, lc = rgbd_cperm(c4 * 4 + e);  // the channel stored at this position
                if (lc < C - C % 16)
                    o[e] = __fmaf_rn(w00, v00[e], __fmaf_rn(w01, v01[e], __fmaf_rn(w11, v11[e], __fmul_rn(w10, v10[e]))));
                else
                    o[e] = __fmaf_rn(w11, v11[e], __fmaf_rn(w10, v10[e], __fmaf_rn(w00, v00[e], __fmul_rn(w01, v01[e]))));
            }
        }
        *reinterpret_cast<f32x4*>(y + ((n * H + oy) * (size_t)W + ox) * cs + c4 * 4) = o;
    }
}

int launch_bilinear(const float* x, int N, int h, int w, int cs, float* y, int H, int W, hipStream_t s, const float* x1, float* y1,
                    int ref_channels)
{
    if (!x1 != !y1) return RGBD_EINVAL;
    const size_t work = (size_t)N * H * W * (cs / 4);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipLaunchKernelGGL(bilinear_kernel, dim3(grid_for(work), x1 ? 2 : 1), dim3(256), 0, s, x, N, h, w, cs, y, H, W, sy, sx, x1, y1,
                       ref_channels > 0 ? 1 + ref_channels : 0);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// ---------------------------------------------------------------------------------------------
// Global average pool per (n, c): fixed reduction order (per-thread strided partial sums, then a fixed LDS tree),
// so the result does not depend on scheduling.  grid = (ceil(C/64), N), block = 256 = 64 channels x 4 pixel lanes.
__global__ void channel_mean_kernel(const float* __restrict__ x, int HW, int cs, int C, float* __restrict__ mean, int mstride)
{
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r = threadIdx.x >> 6;
    const size_t n = blockIdx.y;
    float s = 0.f;
    if (c < C) {
        // same summation order as ever (each of the 4 chains adds its pixels r, r+4, r+8, ... one after the other: the
        // SE gate feeds the entropy parameters, so the order is part of the bitstream contract); the loads of 8 steps are
        // issued together so that a chain is bound by bandwidth, not by one memory round trip per pixel
        const float* b = x + n * (size_t)HW * cs + c;
        const size_t step = (size_t)4 * cs;
        const float* q = b + (size_t)r * cs;
        int p = r;
        for (; p + 28 < HW; p += 32, q += 8 * step) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = q[k * step];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; p < HW; p += 4, q += step) s += *q;
    }
    part[r][threadIdx.x & 63] = s;
    __syncthreads();
    if (r == 0 && c < C) {
        const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        mean[n * mstride + c] = t / (float)HW;
    }
}

int launch_channel_mean(const float* x, int N, int HW, int cs, int C, float* mean, hipStream_t s)
{
    return launch_channel_mean_strided(x, N, HW, cs, C, mean, C, s);
}

// mean[n * mstride + c]: the means of two tensors can land side by side, as if they had been concatenated first
int launch_channel_mean_strided(const float* x, int N, int HW, int cs, int C, float* mean, int mstride, hipStream_t s)
{
    hipLaunchKernelGGL(channel_mean_kernel, dim3((C + 63) / 64, N), dim3(256), 0, s, x, HW, cs, C, mean, mstride);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// SE gate: scale[n][c] = sigmoid(W1 @ relu(W0 @ mean[n])).  W0: [hidden][C]; W1 is stored TRANSPOSED, [hidden][C],
// so both passes read weights with consecutive lanes on consecutive addresses.
// pass 1: one wavefront per (hidden unit, image); lanes stride the dot product, fixed shuffle-tree reduction.
__global__ __launch_bounds__(64) void se_hidden_kernel(const float* __restrict__ mean, int C, int hidden,
                                                       const float* __restrict__ w0, float* __restrict__ hid, int mstride, int perm)
{
    const int j = blockIdx.x;
    const size_t n = blockIdx.y;
    const int lane = threadIdx.x;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(w0[(size_t)j * C + c], mean[n * mstride + rgbd_cperm(c, perm)], s);
#pragma unroll
    for (int off = 32; off; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) hid[n * hidden + j] = fmaxf(s, 0.f);
}

// pass 2: one thread per (image, channel), sequential over the hidden units.
__global__ void se_gate_kernel(const float* __restrict__ hid, int C, int hidden, const float* __restrict__ w1t,
                               float* __restrict__ scale, int perm)
{
    extern __shared__ float hsh[];
    const size_t n = blockIdx.y;
    for (int j = threadIdx.x; j < hidden; j += blockDim.x) hsh[j] = hid[n * hidden + j];
    __syncthreads();
    const int pc = blockIdx.x * blockDim.x + threadIdx.x;  // position in the gate vector (= in the gated tensor)
    if (pc >= C) return;
    const int c = rgbd_cperm(pc, perm);                     // the channel stored there
    float s = 0.f;
    for (int j = 0; j < hidden; ++j) s = fmaf(w1t[(size_t)j * C + c], hsh[j], s);
    scale[n * C + pc] = perm ? rgbd_sigmoid_ref(s) : 1.0f / (1.0f + expf(-s));
}

// (Single-launch variants were built and measured twice.  Round 2: every workgroup recomputing the hidden layer into LDS, then
// gating its channels -- at B = 1 the 4 waves of a workgroup walk 44 hidden units each, one memory round trip after the
// other: ~100 us instead of 2 x 14 us.  Round 4: hidden layer + gate + rescale in one launch behind the means (eight waves,
// two units in flight each; bit-identical): the entropy-parameter nets' SE blocks have 1280 ... 2816 channels, so every one
// of the ~1,000 workgroups that rescale a tensor re-reads up to 2 MB of fc.0 weights -- 62.3 vs 56.6 ms per c3 step on the
// same box.  Each SE stage reduces over a different axis (pixels, channels, hidden units, then back out); fusing
// neighbours trades a ~4 us launch for recomputation that costs more.  The launches below stay.)
int launch_se_fc(const float* mean, int N, int C, int hidden, const float* w0, const float* w1t, float* hid,
                 float* scale, hipStream_t s, int mstride, int perm)
{
    hipLaunchKernelGGL(se_hidden_kernel, dim3(hidden, N), dim3(64), 0, s, mean, C, hidden, w0, hid, mstride > 0 ? mstride : C, perm);
    hipLaunchKernelGGL(se_gate_kernel, dim3((C + 255) / 256, N), dim3(256), (size_t)hidden * sizeof(float), s, hid, C,
                       hidden, w1t, scale, perm);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

__global__ void copy_channels_kernel(const float* __restrict__ src, int scs, float* __restrict__ dst, int dcs,
                                     size_t npix, int c4n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix * c4n; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const size_t pix = i / c4n;
        *reinterpret_cast<f32x4*>(dst + pix * dcs + c4 * 4) = *reinterpret_cast<const f32x4*>(src + pix * scs + c4 * 4);
    }
}

__global__ void channel_scale_to_kernel(const float* __restrict__ x, int HW, int xcs, int sstride,
                                        const float* __restrict__ scale, int mode, float* __restrict__ y, int ycs,
                                        size_t total4, int c4n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const size_t pix = i / c4n;
        const size_t n = pix / HW;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * xcs + c4 * 4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + n * sstride + c4 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = __fmul_rn(v[e], sc[e]);
            o[e] = mode ? __fadd_rn(v[e], t) : t;
        }
        *reinterpret_cast<f32x4*>(y + pix * ycs + c4 * 4) = o;
    }
}

// y[:, :C] = x * s (mode 0) or x + x * s (mode 1) with independent channel strides (C multiple of 4)
int launch_channel_scale_to(const float* x, int N, int HW, int xcs, int C, const float* scale, int mode, float* y, int ycs,
                            hipStream_t s)
{
    return launch_channel_scale_to_strided(x, N, HW, xcs, C, scale, C, mode, y, ycs, s);
}

// scale[n * sstride + c]: the gate of a channel slice of a wider (virtually concatenated) tensor
int launch_channel_scale_to_strided(const float* x, int N, int HW, int xcs, int C, const float* scale, int sstride, int mode,
                                    float* y, int ycs, hipStream_t s)
{
    if (C % 4 || sstride % 4) return RGBD_EINVAL;
    const size_t total4 = (size_t)N * HW * (C / 4);
    hipLaunchKernelGGL(channel_scale_to_kernel, dim3(grid_for(total4)), dim3(256), 0, s, x, HW, xcs, sstride, scale, mode, y,
                       ycs, total4, C / 4);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

int launch_copy_channels(const float* src, int scs, float* dst, int dcs, int npix, int C, hipStream_t s)
{
    if (C % 4 || scs % 4 || dcs % 4) return RGBD_EINVAL;
    const size_t work = (size_t)npix * (C / 4);
    hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(work)), dim3(256), 0, s, src, scs, dst, dcs, (size_t)npix,
                       C / 4);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// First analysis conv (3 / 1 -> N, k 5, stride 2, pad 2): the 25 taps x C real input values of every output pixel, packed
// densely into KP "channels" so that the conv becomes a 1x1 layer with K = KP (80 / 32) instead of 25 taps x 16 padded
// channels.  Term n = tap * C + c sits where the 1x1 kernel's MFMA order (k-step e outer, lane group q inner: packed
// index q * 4 + e) visits it n-th, so every output keeps the fma chain of the tap-by-tap form (taps ascending, channels
// ascending; the padding terms are exact zeros either way).
__global__ void im2col5s2_kernel(const float* __restrict__ x, int H, int W, int cs, int C, float* __restrict__ y, int OH,
                                 int OW, int KP, size_t total4)
{
    const int k4n = KP / 4, nterm = 25 * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int kq = (int)(i % k4n);
        const size_t pix = i / k4n;
        const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH);
        const size_t n = pix / ((size_t)OW * OH);
        const int j = kq >> 2, q = kq & 3;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int term = j * 16 + e * 4 + q;
            if (term < nterm) {
                const int t = term / C, c = term - t * C;
                const int iy = 2 * oy - 2 + t / 5, ix = 2 * ox - 2 + t % 5;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[e] = x[((n * H + iy) * W + ix) * cs + c];
            }
        }
        *reinterpret_cast<f32x4*>(y + pix * KP + kq * 4) = v;
    }
}

int launch_im2col5s2(const float* x, int N, int H, int W, int cs, int C, float* y, int OH, int OW, int KP, hipStream_t s)
{
    if (KP % 16 || 25 * C > KP || C < 1 || C > cs) return RGBD_EINVAL;
    const size_t total4 = (size_t)N * OH * OW * (KP / 4);
    hipLaunchKernelGGL(im2col5s2_kernel, dim3(grid_for(total4)), dim3(256), 0, s, x, H, W, cs, C, y, OH, OW, KP, total4);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// A kernel rather than hipMemsetAsync: these fills sit inside the bodies that are captured into HIP graphs, and a memset
// node of a replayed graph was observed not to clear its buffer (the encoder's error flag came back set with whatever the
// workspace held before) -- a kernel node behaves.
__global__ void fill_zero_kernel(float* __restrict__ p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

int launch_fill_zero(float* p, size_t n, hipStream_t s)
{
    if (!n) return RGBD_OK;
    hipLaunchKernelGGL(fill_zero_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, s, p, n);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// ---------------------------------------------------------------------------------------------
// Small-tensor convolution in the arithmetic of the reference's CPU path for such tensors (torch's im2col + MKL sgemm route:
// batch 1, kernel <= 3, at most 20480 input elements; DESIGN.md 4a, oracle/cpu_arith.c orc_conv_im2col_kblocks): per output
// ONE thread walks k = c * KH * KW + ky * KW + kx in ascending order, a fresh fma chain per K block (kb[0..nb]: boundaries
// in k), out = (S_0 + bias) + S_1 + ...  The layers that take this path are tiny (the ESA pooled branch, latent-grid layers of
// small images): a few MFLOP each, so a plain vector-ALU kernel reading the MFMA kernels' packed weights in place.
// x / y: NHWC, channels permuted (rgbd_cperm); w: packed [cout_pad][ntaps][cin_pad] (both channel axes permuted).
__global__ void small_conv_ref_kernel(SmallConvArgs a)
{
    const size_t total = (size_t)a.N * a.OH * a.OW * a.O;
    const int taps = a.K * a.K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % a.O);
        size_t t = i / a.O;
        const int ox = (int)(t % a.OW);
        t /= a.OW;
        const int oy = (int)(t % a.OH);
        const size_t n = t / a.OH;
        if (a.ckbd && ((oy + ox) & 1) != (a.ckbd == 1 ? 1 : 0)) continue;  // (the other half is left untouched)
        const int po = rgbd_cperm(o);
        const float* wrow = a.w + (size_t)po * taps * a.cin_pad;
        float tot = 0.f;
        for (int b = 0; b < a.nb; ++b) {
            float s = 0.f;
            for (int k = a.kb[b]; k < a.kb[b + 1]; ++k) {
                const int c = k / taps, tp = k - c * taps, ky = tp / a.K, kx = tp - ky * a.K;
                const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
                if (iy < 0 || iy >= a.H || ix < 0 || ix >= a.W) continue;  // (a zero of the im2col matrix: fma(0, w, s) == s)
                const int pc = rgbd_cperm(c);
                s = __fmaf_rn(a.x[((n * a.H + iy) * a.W + ix) * (size_t)a.xcs + pc], wrow[(size_t)tp * a.cin_pad + pc], s);
            }
            tot = b == 0 ? __fadd_rn(s, a.bias[po]) : __fadd_rn(tot, s);
        }
        const size_t opix = (n * a.OH + oy) * a.OW + ox;
        if (a.res1) tot = __fadd_rn(tot, a.res1[opix * a.r1cs + po]);
        if (a.act == ACT_RELU) tot = fmaxf(tot, 0.f);
        else if (a.act == ACT_LEAKY) tot = tot > 0.f ? tot : __fmul_rn(tot, 0.01f);
        else if (a.act == ACT_SIGMOID) tot = rgbd_sigmoid_ref(tot);
        if (a.mul) tot = __fmul_rn(tot, a.mul[opix * a.mcs + po]);
        if (a.res2) tot = __fadd_rn(tot, a.res2[opix * a.r2cs + po]);
        a.y[opix * a.ycs + po] = tot;
        if (a.y2) a.y2[opix * a.y2cs + po] = tot;
    }
}

int launch_small_conv_ref(const SmallConvArgs& a, hipStream_t s)
{
    if (a.nb < 1 || a.nb > 16 || a.K < 1 || a.K > 3) return RGBD_EINVAL;
    const size_t work = (size_t)a.N * a.OH * a.OW * a.O;
    hipLaunchKernelGGL(small_conv_ref_kernel, dim3(grid_for(work)), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// ---------------------------------------------------------------------------------------------
// SE_Block in the reference's CPU arithmetic (oracle/cpu_arith.c orc_mean_row / orc_dot_main / orc_dot_rem).
//
// (1) Global average pool = ATen's cascade sum over the H*W values of a channel, 8-lane vectors in 4 interleaved accumulator
// sets: value p of the channel belongs to vector v = p / 8, lane l = p % 8, accumulator k = v % 4 (for the first
// 4 * (nvec / 4) vectors; the other vectors go to accumulator 0 afterwards); every 16 steps an accumulator is folded into a
// second-level one; then accumulators 1..3 are added to 0, the n % 8 tail values are summed, the 8 lanes added one by one,
// and the sum is divided by n.  One thread per (channel, accumulator, lane): 32 threads per channel, 8 channels per
// workgroup (consecutive threads = consecutive channels of one pixel: 32-byte pieces of the NHWC rows).
__global__ __launch_bounds__(256) void channel_mean_ref_kernel(const float* __restrict__ x, int HW, int cs, int C,
                                                               float* __restrict__ mean, int mstride)
{
    __shared__ float part[32][8];
    const int cl = threadIdx.x & 7, kl = threadIdx.x >> 3;  // channel inside the block; k * 8 + l
    const int k = kl >> 3, l = kl & 7;
    const int pc = blockIdx.x * 8 + cl;                      // channel POSITION (the mean vector is indexed by position, too)
    const size_t n = blockIdx.y;
    const bool live = pc < C;
    const float* b = x + n * (size_t)HW * cs + (live ? pc : 0);
    const int V = HW < 8 ? 1 : 8;
    const int nvec = HW / V, size_ilp = nvec / 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // the cascade's four levels (16 steps each)
    if (l < V) {
        int i = 0;
        for (; i + 16 <= size_ilp;) {
            for (int j = 0; j < 16; ++j, ++i) a0 = __fadd_rn(a0, b[(size_t)((i * 4 + k) * V + l) * cs]);
            a1 = __fadd_rn(a1, a0);
            a0 = 0.f;
            if ((i & 0xF0) == 0) {
                a2 = __fadd_rn(a2, a1);
                a1 = 0.f;
                if ((i & 0xF00) == 0) {
                    a3 = __fadd_rn(a3, a2);
                    a2 = 0.f;
                }
            }
        }
        for (; i < size_ilp; ++i) a0 = __fadd_rn(a0, b[(size_t)((i * 4 + k) * V + l) * cs]);
        a0 = __fadd_rn(__fadd_rn(__fadd_rn(a0, a1), a2), a3);  // acc[0] += acc[1], acc[2], acc[3]
        if (k == 0)
            for (int v = size_ilp * 4; v < nvec; ++v) a0 = __fadd_rn(a0, b[(size_t)(v * V + l) * cs]);
    }
    part[kl][cl] = a0;
    __syncthreads();
    if (kl < 8) {  // lane l of the channel: accumulators 1..3 onto 0
        float p = part[l][cl];
        p = __fadd_rn(p, part[8 + l][cl]);
        p = __fadd_rn(p, part[16 + l][cl]);
        p = __fadd_rn(p, part[24 + l][cl]);
        part[l][cl] = p;
    }
    __syncthreads();
    if (kl == 0 && live) {
        float fin = 0.f;
        for (int p = nvec * V; p < HW; ++p) fin = __fadd_rn(fin, b[(size_t)p * cs]);
        for (int q = 0; q < V; ++q) fin = __fadd_rn(fin, part[q][cl]);
        mean[n * mstride + pc] = __fdiv_rn(fin, (float)HW);
    }
}

int launch_channel_mean_ref(const float* x, int N, int HW, int cs, int C, float* mean, int mstride, hipStream_t s)
{
    hipLaunchKernelGGL(channel_mean_ref_kernel, dim3((C + 7) / 8, N), dim3(256), 0, s, x, HW, cs, C, mean, mstride);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// (2) The two bias-free Linear layers on ONE vector (nn.Linear with batch 1 -> MKL sgemm with n = 1): every output row is a dot
// product in one of three orders (row_class: 0 main, 1 / 2 remainder rows with one / two accumulators; measured per layer
// shape, refarith_tables.json).  One wavefront per output row; lanes 0..31 are the lanes of the 16-lane accumulators.
//   x: input vector; xperm: its entries sit at channel POSITIONS (the mean vector)    act: 1 ReLU, 3 sigmoid
//   yperm: output row j is stored at position rgbd_cperm(j) (the gate vector)
__device__ __forceinline__ float hred16(float v, int lane)
{
    // lanes l + 8, l + 4, l + 2, l + 1 (the order of the CPU's horizontal add); only lane 0's result is used
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) v = __fadd_rn(v, __shfl_down(v, s, 64));
    (void)lane;
    return v;
}

// form >= 0 overrides the per-row classes.  Form 3 = the reference's nn.Linear on a batch of TWO vectors (MKL sgemm with n = 2;
// measured like the batch-1 classes, oracle/cpu_arith.c orc_linear_b2): K < 48: one k-ordered fma chain from 0; otherwise lane l
// of ONE 16-lane accumulator takes elements 16 v + l (the K % 16 tail as one more masked step), and the lanes are reduced as
// q_i = ((a_i + a_{i+4}) + a_{i+8}) + a_{i+12}, (q_0 + q_1) + (q_2 + q_3).
__global__ __launch_bounds__(64) void se_linear_ref_kernel(const float* __restrict__ w, const float* __restrict__ x, int K, int J,
                                                           int xstride, int xperm, const int* __restrict__ row_class, int act,
                                                           float* __restrict__ y, int ystride, int yperm, int form)
{
    const int j = blockIdx.x, lane = threadIdx.x;
    const size_t n = blockIdx.y;
    const float* wr = w + (size_t)j * K;
    const float* xv = x + n * xstride;
    auto X = [&](int k) { return xv[rgbd_cperm(k, xperm)]; };
    const int cls = form >= 0 ? form : (row_class ? row_class[j] : 0);
    float out;
    if (cls == 3) {
        if (K < 48) {
            float acc = 0.f;
            if (lane == 0)
                for (int k = 0; k < K; ++k) acc = __fmaf_rn(wr[k], X(k), acc);
            out = acc;
        } else {
            const int nb = K / 16, nt = K % 16;
            float acc = 0.f;
            if (lane < 16) {
                for (int v = 0; v < nb; ++v) acc = __fmaf_rn(wr[v * 16 + lane], X(v * 16 + lane), acc);
                if (lane < nt) acc = __fmaf_rn(wr[nb * 16 + lane], X(nb * 16 + lane), acc);
            }
            const float a4 = __shfl_down(acc, 4, 64), a8 = __shfl_down(acc, 8, 64), a12 = __shfl_down(acc, 12, 64);
            const float q = __fadd_rn(__fadd_rn(__fadd_rn(acc, a4), a8), a12);  // lanes 0..3: q_i
            const float h = __fadd_rn(q, __shfl_down(q, 1, 64));                // lanes 0, 2: q_0 + q_1, q_2 + q_3
            out = __fadd_rn(h, __shfl_down(h, 2, 64));
        }
    } else if (cls == 0) {
        const int nb = (K - 1) / 16, nt = (K - 1) % 16;
        float acc = 0.f;
        if (lane == 0) acc = __fmul_rn(wr[0], X(0));
        if (lane < 16)
            for (int v = 0; v < nb; ++v) {
                const int k = 1 + v * 16 + lane;
                acc = __fmaf_rn(wr[k], X(k), acc);
            }
        float S = hred16(lane < 16 ? acc : 0.f, lane);
        S = __shfl(S, 0, 64);
        if (nt) {
            float t = 0.f;
            if (lane < nt) {
                const int k = 1 + nb * 16 + lane;
                t = lane == 0 ? __fmaf_rn(wr[k], X(k), S) : __fmul_rn(wr[k], X(k));
            }
            S = hred16(lane < 16 ? t : 0.f, lane);
        }
        out = S;
    } else {
        const int U = cls, step = U * 16, n2 = (K - 1) / step;
        float acc = 0.f;
        if (lane < step)
            for (int q = 0; q < n2; ++q) {
                const int k = 1 + step * q + lane;
                acc = __fmaf_rn(wr[k], X(k), acc);
            }
        if (U == 2) acc = __fadd_rn(acc, __shfl_down(acc, 16, 64));  // A + B, lane by lane
        int pos = 1 + step * n2, rem = K - pos;
        while (rem >= 16) {
            if (lane < 16) acc = __fmaf_rn(wr[pos + lane], X(pos + lane), acc);
            pos += 16;
            rem -= 16;
        }
        if (lane < rem) acc = __fmaf_rn(wr[pos + lane], X(pos + lane), acc);
        const float S = hred16(lane < 16 ? acc : 0.f, lane);
        out = __fadd_rn(S, __fmul_rn(wr[0], X(0)));
    }
    if (lane == 0) {
        if (act == ACT_RELU) out = fmaxf(out, 0.f);
        else if (act == ACT_SIGMOID) out = rgbd_sigmoid_ref(out);
        y[n * ystride + rgbd_cperm(j, yperm)] = out;
    }
}

int launch_se_fc_ref(const float* mean, int N, int C, int hidden, const float* w0, const float* w1, const int* cls0,
                     const int* cls1, float* hid, float* scale, hipStream_t s, int mstride, int form)
{
    // fc.0: [hidden][C] on the means (by position) -> ReLU -> hid[n][hidden];  fc.2: [C][hidden] (NOT transposed) -> sigmoid
    hipLaunchKernelGGL(se_linear_ref_kernel, dim3(hidden, N), dim3(64), 0, s, w0, mean, C, hidden, mstride > 0 ? mstride : C, 1,
                       cls0, ACT_RELU, hid, hidden, 0, form);
    hipLaunchKernelGGL(se_linear_ref_kernel, dim3(C, N), dim3(64), 0, s, w1, hid, hidden, C, hidden, 0, cls1, ACT_SIGMOID, scale,
                       C, 1, form);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// ---------------------------------------------------------------------------------------------
// Stride-2 transposed convolution in the reference's CPU arithmetic (oneDNN brg_deconv; DESIGN.md 4a, oracle/cpu_arith.c
// orc_deconv_s2): per output phase and column class the taps are accumulated tap by tap (all input channels of a tap, then the
// next tap) in chains -- the reduction order of a GEMM whose K axis is (tap, channel).  The engine builds that GEMM per
// (phase, class): this kernel gathers its A matrix, col[pixel][t * cs + c] = x[n][ty + dy_t][j0 + tx + dx_t][c] (zero outside
// the image), the next one its B matrix from the MFMA kernels' packed weights, and the third scatters the result rows to the
// phase's positions of the output.
struct TapList {
    int n;
    int8_t dy[16], dx[16];
    uint8_t slab[16];
};

__global__ void gather_taps_kernel(const float* __restrict__ x, int B, int h, int w, int cs, int j0, int jw, TapList tl,
                                   float* __restrict__ col)
{
    const int c4n = cs / 4;
    const size_t total = (size_t)B * h * jw * tl.n * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        size_t r = i / c4n;
        const int t = (int)(r % tl.n);
        r /= tl.n;
        const int tx = (int)(r % jw);
        r /= jw;
        const int ty = (int)(r % h);
        const size_t n = r / h;
        const int iy = ty + tl.dy[t], ix = j0 + tx + tl.dx[t];
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < h && ix >= 0 && ix < w) v = *reinterpret_cast<const f32x4*>(x + ((n * h + iy) * (size_t)w + ix) * cs + c4 * 4);
        *reinterpret_cast<f32x4*>(col + (((n * h + ty) * (size_t)jw + tx) * tl.n + t) * cs + c4 * 4) = v;
    }
}

__global__ void gather_wslabs_kernel(const float* __restrict__ wp, int cout_pad, int ntaps_total, int cin_pad, TapList tl,
                                     float* __restrict__ wout)
{
    const int c4n = cin_pad / 4;
    const size_t total = (size_t)cout_pad * tl.n * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        size_t r = i / c4n;
        const int t = (int)(r % tl.n);
        const size_t co = r / tl.n;
        *reinterpret_cast<f32x4*>(wout + (co * tl.n + t) * cin_pad + c4 * 4) =
            *reinterpret_cast<const f32x4*>(wp + (co * ntaps_total + tl.slab[t]) * cin_pad + c4 * 4);
    }
}

// dst[n][2 ty + py][2 (j0 + tx) + px][:] = src[(n h + ty) jw + tx][:]
__global__ void scatter_phase_kernel(const float* __restrict__ src, int B, int h, int jw, int scs, int j0, int py, int px,
                                     float* __restrict__ dst, int OW, int dcs, int c4n)
{
    const size_t total = (size_t)B * h * jw * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        size_t r = i / c4n;
        const int tx = (int)(r % jw);
        r /= jw;
        const int ty = (int)(r % h);
        const size_t n = r / h;
        *reinterpret_cast<f32x4*>(dst + ((n * 2 * h + 2 * ty + py) * (size_t)OW + 2 * (j0 + tx) + px) * dcs + c4 * 4) =
            *reinterpret_cast<const f32x4*>(src + ((n * h + ty) * (size_t)jw + tx) * scs + c4 * 4);
    }
}

int launch_gather_taps(const float* x, int B, int h, int w, int cs, int j0, int jw, int ntap, const int* dy, const int* dx,
                       float* col, hipStream_t s)
{
    if (ntap < 1 || ntap > 16 || cs % 4) return RGBD_EINVAL;
    TapList tl{};
    tl.n = ntap;
    for (int t = 0; t < ntap; ++t) tl.dy[t] = (int8_t)dy[t], tl.dx[t] = (int8_t)dx[t];
    const size_t work = (size_t)B * h * jw * ntap * (cs / 4);
    hipLaunchKernelGGL(gather_taps_kernel, dim3(grid_for(work)), dim3(256), 0, s, x, B, h, w, cs, j0, jw, tl, col);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

int launch_gather_wslabs(const float* wp, int cout_pad, int ntaps_total, int cin_pad, int ntap, const int* slab, float* wout,
                         hipStream_t s)
{
    if (ntap < 1 || ntap > 16 || cin_pad % 4) return RGBD_EINVAL;
    TapList tl{};
    tl.n = ntap;
    for (int t = 0; t < ntap; ++t) tl.slab[t] = (uint8_t)slab[t];
    const size_t work = (size_t)cout_pad * ntap * (cin_pad / 4);
    hipLaunchKernelGGL(gather_wslabs_kernel, dim3(grid_for(work)), dim3(256), 0, s, wp, cout_pad, ntaps_total, cin_pad, tl, wout);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

int launch_scatter_phase(const float* src, int B, int h, int jw, int scs, int j0, int py, int px, float* dst, int OW, int dcs,
                         int C, hipStream_t s)
{
    if (C % 4 || scs % 4 || dcs % 4) return RGBD_EINVAL;
    const size_t work = (size_t)B * h * jw * (C / 4);
    hipLaunchKernelGGL(scatter_phase_kernel, dim3(grid_for(work)), dim3(256), 0, s, src, B, h, jw, scs, j0, py, px, dst, OW, dcs, C / 4);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}


// ---- sigmoid gate of a tensor whose reference sigmoid has scalar tails ----------------------------------------
// y = mul * sigmoid(t) + res2, element by element over NHWC tensors whose channels are stored permuted (rgbd_cperm): what the
// conv epilogue does for ACT_SIGMOID + mul + res2 (layers.py:198-213 a * sigmoid(b) + x; attention.py:84-97 x * sigmoid(.)),
// except that the sigmoid of element (n, c, h, w) is the one torch's CPU kernel applies at flat index i of the reference's
// contiguous [batch, C, H, W] tensor: the Sleef vector form, or -- in the last len % 32 elements of each parallel chunk --
// the scalar form on the C library's expf (exact_math.h).  per_image: the reference codes image by image (batch 1).
__global__ void sigmoid_gate_ref_kernel(const float* __restrict__ t, int tcs, const float* __restrict__ mul, int mcs,
                                        const float* __restrict__ res2, int r2cs, float* __restrict__ y, int ycs, int N, int HW,
                                        int C, int per_image, int threads)
{
    const int C16 = (C + 15) & ~15;
    const size_t total = (size_t)N * HW * C16;
    const long numel = (long)(per_image ? 1 : N) * C * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int sc = (int)(i % C16);       // stored channel
        const size_t pix = i / C16;          // n * HW + hw
        const int c = rgbd_cperm(sc);        // logical channel (the permutation is an involution)
        if (c >= C) {                        // pad channels hold zeros
            y[pix * ycs + sc] = 0.f;
            continue;
        }
        const long n = (long)(pix / HW), hw = (long)(pix % HW);
        const long flat = ((per_image ? 0 : n) * C + c) * HW + hw;
        const float v = t[pix * tcs + sc];
        float g = rgbd_aten_scalar_tail(flat, numel, threads) ? rgbd_sigmoid_ref_scalar(v) : rgbd_sigmoid_ref(v);
        if (mul) g = __fmul_rn(g, mul[pix * mcs + sc]);
        if (res2) g = __fadd_rn(g, res2[pix * r2cs + sc]);
        y[pix * ycs + sc] = g;
    }
}

int launch_sigmoid_gate_ref(const float* t, int tcs, const float* mul, int mcs, const float* res2, int r2cs, float* y, int ycs,
                            int N, int HW, int C, int per_image, int threads, hipStream_t s)
{
    if (!t || !y || N <= 0 || HW <= 0 || C <= 0) return RGBD_EINVAL;
    const size_t total = (size_t)N * HW * ((C + 15) & ~15);
    hipLaunchKernelGGL(sigmoid_gate_ref_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, s, t, tcs,
                       mul, mcs, res2, r2cs, y, ycs, N, HW, C, per_image, threads);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}
