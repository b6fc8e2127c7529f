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
// blockIdx.y == 1: the second operand set (the other modality's tensor of the same shape with its own weights).
struct LnSet {
    const float* x;
    const float* w;
    const float* b;
    float* y;
};
__global__ __launch_bounds__(256) void layernorm_kernel(LnSet s0, LnSet s1, size_t ntok, int C, int xcs, int ycs)
{
    const LnSet st = blockIdx.y ? s1 : s0;
    const float* __restrict__ x = st.x;
    const float* __restrict__ w = st.w;
    const float* __restrict__ b = st.b;
    float* __restrict__ y = st.y;
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
        // the lane's sum of squared deviations: d0^2 fused onto the rounded d1^2, then the other terms in order.  (That is how
        // the compiler contracted `sq += d * d` over the unrolled loop in rounds 1-3 -- `0 + d0 * d0 + d1 * d1` has two legal
        // contractions -- and every STF_united stream since was coded with it; written out so that it cannot change again.
        // Round 4 first spelled the other one, fma(d1, d1, d0 * d0): a last-bit change of the variance that moved the
        // stf_c5 golden's stream lengths by 16 and 12 bytes and was found when its floors were re-recorded.)
        float dv[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int c = lane + 64 * k;
            dv[k] = c < C ? v[k] - mean : 0.f;
        }
        float sq = __fmaf_rn(dv[0], dv[0], __fmul_rn(dv[1], dv[1]));
#pragma unroll
        for (int k = 2; k < 24; ++k) sq = __fmaf_rn(dv[k], dv[k], sq);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.0f / sqrtf(sq / (float)C + 1e-5f);
        float* yp = y + t * ycs;
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int c = lane + 64 * k;
            if (c < C) yp[c] = __fmaf_rn(__fmul_rn(v[k] - mean, rstd), w[c], b[c]);
            else if (c < ycs) yp[c] = 0.f;
        }
    }
}

// The same LayerNorm with 16 lanes per token and one float4 per lane and pass of 64 channels (lane i of a token's 16 holds
// channels 64 r + 4 i .. + 3 in pass r): a wave covers four tokens with 256-byte contiguous segments per load instead of one
// token with 4-byte lanes, and runs ceil(C / 64) passes instead of 24 predicated ones -- the one-wave-per-token form above
// reached 0.7 TB/s on STF_united's 48- and 96-channel stages (profiles/r04_c5_stf_4x512x512_w1_summary.txt: the largest
// kernel of config 5 after the coder).  SAME sums: the leaf of channel c0 < 64 is x[c0] + x[c0 + 64] + ... in order, and the
// xor-butterfly over lanes 32, 16, 8, 4, 2, 1 of the form above is, in this layout, lanes ^8, ^4, ^2, ^1 and then the
// components (0 + 2) + (1 + 3) -- bit-identical (tests/test_gpu_stf.py::test_layernorm_forms_same_bits).  C % 4 == 0.
template <int R>
__global__ __launch_bounds__(256) void layernorm4_kernel(LnSet s0, LnSet s1, size_t ntok, int C, int xcs, int ycs)
{
    const LnSet st = blockIdx.y ? s1 : s0;
    const float* __restrict__ x = st.x;
    const float* __restrict__ w = st.w;
    const float* __restrict__ b = st.b;
    float* __restrict__ y = st.y;
    const int sub = threadIdx.x & 15;
    const size_t grp = (blockIdx.x * (size_t)256 + threadIdx.x) >> 4;
    const size_t ngrp = (size_t)gridDim.x * 16;
    const float fc = (float)C;
    for (size_t t = grp; t < ntok; t += ngrp) {
        const float* xp = x + t * xcs;
        f32x4 v[R];
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int c = 64 * r + 4 * sub;
            v[r] = c < C ? *reinterpret_cast<const f32x4*>(xp + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += v[r][j];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += __shfl_xor(s[j], o);
        const float mean = ((s[0] + s[2]) + (s[1] + s[3])) / fc;
        float q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // the same leaf as above: d0^2 fused onto the rounded d1^2 (0 for a single pass), then in order
            const float d0 = 4 * sub < C ? v[0][j] - mean : 0.f;
            const float d1 = (R > 1 && 64 + 4 * sub < C) ? v[R > 1 ? 1 : 0][j] - mean : 0.f;
            q[j] = __fmaf_rn(d0, d0, __fmul_rn(d1, d1));
        }
#pragma unroll
        for (int r = 2; r < R; ++r) {
            const int c = 64 * r + 4 * sub;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = c < C ? v[r][j] - mean : 0.f;
                q[j] = __fmaf_rn(d, d, q[j]);
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = __fadd_rn(q[j], __shfl_xor(q[j], o));
        const float sq = __fadd_rn(__fadd_rn(q[0], q[2]), __fadd_rn(q[1], q[3]));
        const float rstd = 1.0f / sqrtf(sq / fc + 1e-5f);
        float* yp = y + t * ycs;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int c = 64 * r + 4 * sub;
            if (c < C) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c), bv = *reinterpret_cast<const f32x4*>(b + c);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = __fmaf_rn(__fmul_rn(v[r][j] - mean, rstd), wv[j], bv[j]);
                *reinterpret_cast<f32x4*>(yp + c) = o;
            } else if (c < ycs) {
                *reinterpret_cast<f32x4*>(yp + c) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

static int g_ln_form = -1;  // -1: by shape; 0: one wave per token; 1: 16 lanes per token (tests)
extern "C" void rgbd_debug_force_layernorm_form(int32_t form) { g_ln_form = form; }

int launch_layernorm(const float* x, size_t ntok, int C, int xcs, const float* w, const float* b, float* y, int ycs,
                     hipStream_t s, const float* x1, const float* w1, const float* b1, float* y1)
{
    if (C > 1536 || ycs > 1536 || C <= 0) return RGBD_EINVAL;
    const LnSet s0{x, w, b, y}, s1{x1, w1, b1, y1};
    const auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool vec_ok = C % 4 == 0 && xcs % 4 == 0 && ycs % 4 == 0 && al16(x) && al16(w) && al16(b) && al16(y) &&
                        (!x1 || (al16(x1) && al16(w1) && al16(b1) && al16(y1)));
    if (g_ln_form == 1 && !vec_ok) return RGBD_EINVAL;
    if (vec_ok && g_ln_form != 0) {
        const int passes = (std::max(C, ycs) + 63) / 64;  // (the zero fill of y's pad channels rides on the passes too)
        size_t g = (ntok + 15) / 16;
        if (g > 8192) g = 8192;
        const dim3 grid((unsigned)g, x1 ? 2 : 1);
#define RGBD_LN4(R_)                                                                                          \
    if (passes <= R_) {                                                                                       \
        hipLaunchKernelGGL(layernorm4_kernel<R_>, grid, dim3(256), 0, s, s0, s1, ntok, C, xcs, ycs);          \
        HIP_TRY(hipGetLastError());                                                                           \
        return RGBD_OK;                                                                                       \
    }
        RGBD_LN4(1) RGBD_LN4(2) RGBD_LN4(3) RGBD_LN4(4) RGBD_LN4(6) RGBD_LN4(8) RGBD_LN4(12) RGBD_LN4(16) RGBD_LN4(24)
#undef RGBD_LN4
    }
    size_t g = (ntok + 3) / 4;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)g, x1 ? 2 : 1), dim3(256), 0, s, s0, s1, ntok, C, xcs, ycs);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// Stand-alone LayerNorm over the last dimension of a [ntok][xcs] fp32 device tensor (nn.LayerNorm(C, eps = 1e-5),
// stf_united.py:143,155,225,263,387-391) -- the operator boundary the tests use.
extern "C" int rgbd_layernorm(const float* x, int64_t ntok, int32_t C, int32_t xcs, const float* w, const float* b, float* y,
                              int32_t ycs, void* stream)
{
    if (!x || !w || !b || !y || ntok <= 0 || xcs < C || ycs < C) return RGBD_EINVAL;
    return launch_layernorm(x, (size_t)ntok, C, xcs, w, b, y, ycs, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// WindowAttention + the window partition / cyclic shift around it (stf_united.py:48-114, 177-203, 329-352) on the matrix
// cores.  qkv: [B,H,W,3C] with channel = which*C + head*16 + d (the layout nn.Linear(dim, 3*dim) produces); out: [B,H,W,C].
// A 4x4 window with head_dim 16 is two 16x16x16 products per (window, head): S = (q * 0.25) k^T and O = softmax(S + bias
// [+ mask]) v -- four v_mfma_f32_16x16x4_f32 each.  One wavefront walks WA_PAIRS (window, head) pairs, four wavefronts per
// workgroup.  Operand layouts of the instruction (A: lane l = row l % 16, k = l / 16; B: col l % 16, k = l / 16; D: rows
// 4 (l / 16) + r in register r, col l % 16):
//   S:  A = q[token i = l % 16][d = 4 s + l / 16], B = k[token j = l % 16][d = 4 s + l / 16]  (k-step s = 0..3), so every score
//       is ONE fused-multiply-add chain over d = 0 .. 15 in order, starting from 0;
//   O:  A = P[i][j = 4 s + l / 16] (P goes through LDS: it comes out of the first product in the D layout), B = v[token
//       j = 4 s + l / 16][dim l % 16] read straight from qkv; the chain over j = 0 .. 15 in order.
// The relative position bias and the shifted-window mask (-100 between different regions of the rolled frame) are added to
// the accumulator registers, the softmax runs over the 16 lanes that hold a row (sum: four consecutive columns in order,
// then the groups pairwise -- a fixed tree).  Against the vector-ALU form of rounds 1-3 (one wavefront per pair; the
// compiler had turned its dot products into packed multiplies followed by separate adds, two roundings per term) the
// results differ in the last bits -- fewer roundings here; same-box A/B on the STF golden: |dPSNR| vs the reference 1.05e-4 /
// 2.4e-5 dB before, 2.7e-5 / 8.1e-5 dB now.
// shift > 0: the window lives in the rolled frame (token (ys,xs) is pixel ((ys+shift)%H, (xs+shift)%W)).
// blockIdx.y == 1: the second operand set (the other modality).
#define WA_PAIRS 4
struct WaSet {
    const float* qkv;
    const float* rpb;  // [49][heads]
    float* out;
};
__global__ __launch_bounds__(256) void window_attention_kernel(WaSet s0, WaSet s1, int B, int H, int W, int C, int qcs, int heads,
                                                               int shift, int ocs, int npairs)
{
    __shared__ float sq[4][16][17], sk[4][16][17], sp[4][16][17];
    const WaSet st = blockIdx.y ? s1 : s0;
    const float* __restrict__ qkv = st.qkv;
    const float* __restrict__ rpb = st.rpb;
    float* __restrict__ out = st.out;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int nwx = W / 4, nwy = H / 4;
    for (int it = 0; it < WA_PAIRS; ++it) {
        const int pair = (blockIdx.x * 4 + wv) * WA_PAIRS + it;  // wave-uniform
        if (pair >= npairs) break;
        const int head = pair % heads;
        const int win = pair / heads;
        const int wj = win % nwx, wi = (win / nwx) % nwy, b = win / (nwx * nwy);
        const size_t img = (size_t)b * H * W;
        // stage q (scaled) and k of this pair: lane loads token t = lane / 4, dims 4 (lane % 4) .. + 3 (64-byte rows)
        {
            const int t = lane >> 2, d0 = (lane & 3) * 4;
            const int ys = wi * 4 + (t >> 2), xs = wj * 4 + (t & 3);
            const int y = (ys + shift) % H, x = (xs + shift) % W;
            const float* p = qkv + (img + (size_t)y * W + x) * qcs + head * 16 + d0;
            const f32x4 qv = *reinterpret_cast<const f32x4*>(p);
            const f32x4 kv = *reinterpret_cast<const f32x4*>(p + C);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sq[wv][t][d0 + e] = qv[e] * 0.25f;  // head_dim ** -0.5, head_dim = 16
                sk[wv][t][d0 + e] = kv[e];
            }
        }
        // v operands straight from memory: lane (q, c = l15), k-step s: v[token 4 s + q][dim c]
        float vb[4];
#pragma unroll
        for (int sstep = 0; sstep < 4; ++sstep) {
            const int t = 4 * sstep + q;
            const int ys = wi * 4 + (t >> 2), xs = wj * 4 + (t & 3);
            const int y = (ys + shift) % H, x = (xs + shift) % W;
            vb[sstep] = qkv[(img + (size_t)y * W + x) * qcs + 2 * C + head * 16 + l15];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes have landed (its own region only)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sstep = 0; sstep < 4; ++sstep)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sq[wv][l15][4 * sstep + q], sk[wv][l15][4 * sstep + q], acc, 0, 0, 0);
        // acc[r] = S[i = 4 q + r][j = l15]
        const int j = l15, jy = j >> 2, jx = j & 3;
        int reg_j = 0;
        if (shift > 0) {
            const int ys = wi * 4 + jy, xs = wj * 4 + jx;
            reg_j = (ys < H - 4 ? 0 : (ys < H - shift ? 1 : 2)) * 3 + (xs < W - 4 ? 0 : (xs < W - shift ? 1 : 2));
        }
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r, iy = i >> 2, ix = i & 3;
            float sc = acc[r] + rpb[((iy - jy + 3) * 7 + (ix - jx + 3)) * heads + head];  // relative_position_index, stf_united.py:62-72
            if (shift > 0) {
                const int ys = wi * 4 + iy, xs = wj * 4 + ix;
                const int reg_i = (ys < H - 4 ? 0 : (ys < H - shift ? 1 : 2)) * 3 + (xs < W - 4 ? 0 : (xs < W - shift ? 1 : 2));
                if (reg_j != reg_i) sc += -100.0f;
            }
            // softmax over the row: the 16 lanes of this lane group hold its 16 columns
            float mx = sc;
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            mx = fmaxf(mx, __shfl_xor(mx, 2));
            mx = fmaxf(mx, __shfl_xor(mx, 4));
            mx = fmaxf(mx, __shfl_xor(mx, 8));
            const float ex = expf(sc - mx);
            // sum in the scalar form's order: ((e0 + e1) + e2) + e3 inside each group of four columns, then the groups pairwise
            float sum = ex + __shfl_down(ex, 1);
            sum += __shfl_down(ex, 2);
            sum += __shfl_down(ex, 3);
            sum = __shfl(sum, lane & ~3);  // every lane of the group of four takes the group's sum (held by its first lane)
            sum += __shfl_xor(sum, 4);
            sum += __shfl_xor(sum, 8);
            pr[r] = ex / sum;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sp[wv][4 * q + r][l15] = pr[r];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sstep = 0; sstep < 4; ++sstep)
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(sp[wv][l15][4 * sstep + q], vb[sstep], o, 0, 0, 0);
        // o[r] = O[token 4 q + r][dim l15]: 64-byte rows per store instruction
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 4 * q + r;
            const int ys = wi * 4 + (t >> 2), xs = wj * 4 + (t & 3);
            const int y = (ys + shift) % H, x = (xs + shift) % W;
            out[(img + (size_t)y * W + x) * ocs + head * 16 + l15] = o[r];
        }
        __builtin_amdgcn_wave_barrier();  // the next pair overwrites this wave's LDS region
    }
}

int launch_window_attention(const float* qkv, int B, int H, int W, int C, int qcs, int heads, int shift, const float* rpb,
                            float* out, int ocs, hipStream_t s, const float* qkv1, const float* rpb1, float* out1)
{
    if (H % 4 || W % 4 || C != heads * 16 || shift < 0 || shift >= 4) return RGBD_EINVAL;  // window 4, head_dim 16
    const size_t pairs = (size_t)B * (H / 4) * (W / 4) * heads;
    if (pairs > 0x7fffffffu) return RGBD_EINVAL;
    const size_t blocks = (pairs + 4 * WA_PAIRS - 1) / (4 * WA_PAIRS);
    const WaSet s0{qkv, rpb, out}, s1{qkv1, rpb1, out1};
    hipLaunchKernelGGL(window_attention_kernel, dim3((unsigned)blocks, qkv1 ? 2 : 1), dim3(256), 0, s, s0, s1, B, H, W, C, qcs, heads,
                       shift, ocs, (int)pairs);
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
