// Swin-transformer pieces of STF_united (reference models/stf_united.py) on gfx950, all on NHWC fp32 tensors with channel
// stride cs (multiple of 16): LayerNorm over the channels of a token, 4x4 window attention with relative position
// bias and the shifted-window mask, patch merging (2x2 gather) and PixelShuffle.  The linear layers around them are
// 1x1 convolutions of conv_mfma.hip (GELU and the residual adds live in their epilogues).
// These kernels are HBM/latency bound: one wavefront per token (LayerNorm) or per (window, head) (attention).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// nn.LayerNorm(C, eps=1e-5) per token (stf_united.py:143,155,225,263,387-391): biased variance, two-pass in registers.
// One wavefront per token; lane l owns channels l, l+64, ...  (C <= 1536).  Pad channels of y are zeroed.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, size_t ntok, int C, int xcs,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ y, int ycs)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t t = wave; t < ntok; t += nwaves) {
        const float* xp = x + t * xcs;
        float v[24];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int c = lane + 64 * k;
            v[k] = c < C ? xp[c] : 0.f;
            sum += v[k];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int c = lane + 64 * k;
            const float d = c < C ? v[k] - mean : 0.f;
            sq += d * d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.0f / sqrtf(sq / (float)C + 1e-5f);
        float* yp = y + t * ycs;
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int c = lane + 64 * k;
            if (c < C) yp[c] = (v[k] - mean) * rstd * w[c] + b[c];
            else if (c < ycs) yp[c] = 0.f;
        }
    }
}

int launch_layernorm(const float* x, size_t ntok, int C, int xcs, const float* w, const float* b, float* y, int ycs,
                     hipStream_t s)
{
    if (C > 1536 || ycs > 1536 || C <= 0) return RGBD_EINVAL;
    size_t g = (ntok + 3) / 4;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)g), dim3(256), 0, s, x, ntok, C, xcs, w, b, y, ycs);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// ---------------------------------------------------------------------------------------------
// WindowAttention + the window partition / cyclic shift around it (stf_united.py:48-114, 177-203, 329-352).
// qkv: [B,H,W,3C] with channel = which*C + head*16 + d (the layout nn.Linear(dim, 3*dim) produces); out: [B,H,W,C].
// One wavefront per (window, head): 16 tokens x 16 dims.  Lane (i = l/4, g = l%4) owns scores (i, 4g..4g+3) and outputs
// (i, 4g..4g+3).  shift > 0: the window lives in the rolled frame (token (ys,xs) is pixel ((ys+shift)%H, (xs+shift)%W))
// and pairs from different mask regions get -100 before the softmax.
__global__ __launch_bounds__(64) void window_attention_kernel(const float* __restrict__ qkv, int B, int H, int W, int C,
                                                              int qcs, int heads, int shift,
                                                              const float* __restrict__ rpb,  // [49][heads]
                                                              float* __restrict__ out, int ocs)
{
    __shared__ float sq[16][17], sk[16][17], sv[16][17], sp[16][17];
    const int lane = threadIdx.x;
    const int head = blockIdx.x % heads;
    const int win = blockIdx.x / heads;
    const int nwx = W / 4, nwy = H / 4;
    const int wj = win % nwx, wi = (win / nwx) % nwy, b = win / (nwx * nwy);
    // stage q (scaled), k, v: lane loads token t = lane/4, dims 4*(lane%4) .. +3
    {
        const int t = lane >> 2, d0 = (lane & 3) * 4;
        const int ys = wi * 4 + (t >> 2), xs = wj * 4 + (t & 3);
        const int y = (ys + shift) % H, x = (xs + shift) % W;
        const float* p = qkv + (((size_t)b * H + y) * W + x) * qcs + head * 16 + d0;
        const f32x4 q = *reinterpret_cast<const f32x4*>(p);
        const f32x4 k = *reinterpret_cast<const f32x4*>(p + C);
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 2 * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sq[t][d0 + e] = q[e] * 0.25f;  // head_dim ** -0.5, head_dim = 16
            sk[t][d0 + e] = k[e];
            sv[t][d0 + e] = v[e];
        }
    }
    __syncthreads();
    const int i = lane >> 2, g = lane & 3;
    const int iy = i >> 2, ix = i & 3;
    float sc[4];
    int reg_i = 0;
    if (shift > 0) {
        const int ys = wi * 4 + iy, xs = wj * 4 + ix;
        reg_i = (ys < H - 4 ? 0 : (ys < H - shift ? 1 : 2)) * 3 + (xs < W - 4 ? 0 : (xs < W - shift ? 1 : 2));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = g * 4 + e;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) acc += sq[i][d] * sk[j][d];
        const int jy = j >> 2, jx = j & 3;
        acc += rpb[((iy - jy + 3) * 7 + (ix - jx + 3)) * heads + head];  // relative_position_index, stf_united.py:62-72
        if (shift > 0) {
            const int ys = wi * 4 + jy, xs = wj * 4 + jx;
            const int reg_j = (ys < H - 4 ? 0 : (ys < H - shift ? 1 : 2)) * 3 + (xs < W - 4 ? 0 : (xs < W - shift ? 1 : 2));
            if (reg_j != reg_i) acc += -100.0f;
        }
        sc[e] = acc;
    }
    float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sc[e] = expf(sc[e] - mx);
        sum += sc[e];
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) sp[i][g * 4 + e] = sc[e] / sum;
    __syncthreads();
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float pj = sp[i][j];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += pj * sv[j][g * 4 + e];
    }
    {
        const int ys = wi * 4 + iy, xs = wj * 4 + ix;
        const int y = (ys + shift) % H, x = (xs + shift) % W;
        *reinterpret_cast<f32x4*>(out + (((size_t)b * H + y) * W + x) * ocs + head * 16 + g * 4) = o;
    }
}

int launch_window_attention(const float* qkv, int B, int H, int W, int C, int qcs, int heads, int shift, const float* rpb,
                            float* out, int ocs, hipStream_t s)
{
    if (H % 4 || W % 4 || C != heads * 16 || shift < 0 || shift >= 4) return RGBD_EINVAL;  // window 4, head_dim 16
    const size_t blocks = (size_t)B * (H / 4) * (W / 4) * heads;
    if (blocks > 0x7fffffffu) return RGBD_EINVAL;
    hipLaunchKernelGGL(window_attention_kernel, dim3((unsigned)blocks), dim3(64), 0, s, qkv, B, H, W, C, qcs, heads, shift,
                       rpb, out, ocs);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// ---------------------------------------------------------------------------------------------
// PatchMerging gather (stf_united.py:240-245): y[b,h,w] = cat(x[2h,2w], x[2h+1,2w], x[2h,2w+1], x[2h+1,2w+1]) on channels
__global__ void patch_merge_gather_kernel(const float* __restrict__ x, int B, int H, int W, int C, int xcs,
                                          float* __restrict__ y, int ycs)
{
    const int OH = H / 2, OW = W / 2, c4n = C / 4;
    const size_t total = (size_t)B * OH * OW * 4 * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int q = (int)((i / c4n) % 4);
        const size_t pix = i / ((size_t)4 * c4n);
        const int ow = (int)(pix % OW), oh = (int)((pix / OW) % OH);
        const size_t b = pix / ((size_t)OW * OH);
        const int ih = 2 * oh + (q & 1), iw = 2 * ow + (q >> 1);
        *reinterpret_cast<f32x4*>(y + pix * ycs + q * C + c4 * 4) =
            *reinterpret_cast<const f32x4*>(x + ((b * H + ih) * W + iw) * xcs + c4 * 4);
    }
}

int launch_patch_merge_gather(const float* x, int B, int H, int W, int C, int xcs, float* y, int ycs, hipStream_t s)
{
    if (H % 2 || W % 2 || C % 4) return RGBD_EINVAL;
    const size_t total = (size_t)B * (H / 2) * (W / 2) * C;
    size_t g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(patch_merge_gather_kernel, dim3((unsigned)g), dim3(256), 0, s, x, B, H, W, C, xcs, y, ycs);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// ---------------------------------------------------------------------------------------------
// nn.PixelShuffle(2) on NHWC: y[b, 2h+i, 2w+j, c] = x[b, h, w, 4c + 2i + j]   (stf_united.py:257-266, 550-558)
__global__ void pixel_shuffle2_kernel(const float* __restrict__ x, int B, int H, int W, int Co, int xcs,
                                      float* __restrict__ y, int ycs)
{
    const size_t total = (size_t)B * 2 * H * 2 * W * ycs;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ycs);
        const size_t pix = i / ycs;
        const int ow = (int)(pix % (2 * W)), oh = (int)((pix / (2 * W)) % (2 * H));
        const size_t b = pix / ((size_t)4 * W * H);
        float v = 0.f;
        if (c < Co) v = x[((b * H + (oh >> 1)) * W + (ow >> 1)) * xcs + 4 * c + 2 * (oh & 1) + (ow & 1)];
        y[i] = v;
    }
}

int launch_pixel_shuffle2(const float* x, int B, int H, int W, int Co, int xcs, float* y, int ycs, hipStream_t s)
{
    const size_t total = (size_t)B * 4 * H * W * ycs;
    size_t g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3((unsigned)g), dim3(256), 0, s, x, B, H, W, Co, xcs, y, ycs);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}
