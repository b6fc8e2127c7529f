/* oracle/cpu_arith.c -- plain-C restatement of the THIRD-PARTY fp32 arithmetic the reference's float path ends in.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
 *
 * The reference (models/elic_united.py, modules/transform/*.py, modules/layers/conv.py:7-34) is PyTorch code; the numbers
 * its convolutions, sigmoids, means and Linear layers produce are decided by libraries that are NOT part of /root/reference:
 *   torch 2.10.0 (CPU, AVX-512 build)  ->  oneDNN 3.7.1 (convolution / deconvolution primitives jit:avx512_core,
 *   jit_1x1:avx512_core, brg_deconv), Intel MKL 2024.2 (sgemm behind nn.Linear and behind the small-tensor conv path),
 *   Sleef (vector expf behind torch.sigmoid), ATen's cascade sum (mean).
 * None of them is vendored, so their *published behaviour* is restated here -- the order in which each output's products
 * are accumulated -- and pinned two ways: (1) tools/refarith/discover.py probes the installed libraries as black boxes in
 * the survey container (absorbing / cancelling probe values reveal the summation tree of every output) and
 * tests/test_oracle_arith.py checks this file against torch bit for bit there; (2) with these functions in place of
 * torch's, oracle/elic_oracle.py still reproduces the reference's golden streams (tests/test_oracle_model.py).
 *
 * What was found (8 threads, Xeon with AVX-512, the machine that produced tests/golden/):
 *   * conv2d, kernel > 1 (jit:avx512_core): the input channels are walked in blocks of 16; inside a block one fma chain
 *     per output runs kh -> kw -> channel, starting from 0; out = (S_0 + bias) + S_1 + S_2 ... in block order.
 *   * conv2d, 1x1 (jit_1x1:avx512_core): the same with larger "reduce blocks" chosen per layer shape (e.g. 192 channels at
 *     64x80: 96 + 96; 1280 channels at 32x40: 384 + 384 + 384 + 128) and the first chain starting from the bias.
 *   * conv_transpose2d, stride 1 (conv:any+jit:avx512_core): blocks of 16 input channels, kh -> kw -> channel, bias last.
 *   * small tensors (batch 1, kernel <= 3, <= 20480 input elements) take ATen's im2col + MKL sgemm path: k runs
 *     channel -> kh -> kw, cut into blocks that MKL chooses from K; out = (S_0 + bias) + S_1 ...
 * Which blocks apply to which layer shape is data (learning-based-rgb-d-image-compression_amd/refarith_tables.json),
 * measured by the probe tool; this file only executes a given structure.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* x [N][C][H][W]; w [O][C][KH][KW]; y [N][O][OH][OW].  bnd[0..nb]: channel boundaries of the blocks.
 * order 0: inside a block 16-channel chunk -> ky -> kx -> c (oneDNN direct / 1x1);  order 1: c -> ky -> kx (im2col + sgemm)
 * bias_mode 0: sum, then + bias; 1: (S_0 + bias) + S_1 ...; 2: the first chain starts from the bias */
void orc_conv_blocks(const float* x, int N, int C, int H, int W, const float* w, int O, int KH, int KW, const float* b,
                     int stride, int pad, const int* bnd, int nb, int order, int bias_mode, float* y, int OH, int OW)
{
#pragma omp parallel
    {
        float* tot = (float*)malloc(sizeof(float) * (size_t)OW * 2);
        float* s = tot + OW;
#pragma omp for collapse(3) schedule(static)
        for (int n = 0; n < N; n++)
            for (int o = 0; o < O; o++)
                for (int oy = 0; oy < OH; oy++) {
                    for (int bi = 0; bi < nb; bi++) {
                        const int c0 = bnd[bi], c1 = bnd[bi + 1];
                        const float init = (bi == 0 && bias_mode == 2 && b) ? b[o] : 0.f;
                        for (int ox = 0; ox < OW; ox++) s[ox] = init;
                        if (order == 0) {
                            for (int cb = c0; cb < c1; cb += 16)
                                for (int ky = 0; ky < KH; ky++) {
                                    const int iy = oy * stride - pad + ky;
                                    if (iy < 0 || iy >= H) continue;
                                    for (int kx = 0; kx < KW; kx++) {
                                        int lo = 0, hi = OW;
                                        while (lo < OW && lo * stride - pad + kx < 0) lo++;
                                        while (hi > lo && (hi - 1) * stride - pad + kx >= W) hi--;
                                        for (int c = cb; c < cb + 16 && c < c1; c++) {
                                            const float wv = w[(((size_t)o * C + c) * KH + ky) * KW + kx];
                                            const float* xr = x + (((size_t)n * C + c) * H + iy) * W - pad + kx;
                                            for (int ox = lo; ox < hi; ox++) s[ox] = fmaf(xr[ox * stride], wv, s[ox]);
                                        }
                                    }
                                }
                        } else {
                            for (int c = c0; c < c1; c++)
                                for (int ky = 0; ky < KH; ky++) {
                                    const int iy = oy * stride - pad + ky;
                                    if (iy < 0 || iy >= H) continue;  /* (a zero in the im2col matrix: fma(0, w, s) == s) */
                                    for (int kx = 0; kx < KW; kx++) {
                                        int lo = 0, hi = OW;
                                        while (lo < OW && lo * stride - pad + kx < 0) lo++;
                                        while (hi > lo && (hi - 1) * stride - pad + kx >= W) hi--;
                                        const float wv = w[(((size_t)o * C + c) * KH + ky) * KW + kx];
                                        const float* xr = x + (((size_t)n * C + c) * H + iy) * W - pad + kx;
                                        for (int ox = lo; ox < hi; ox++) s[ox] = fmaf(xr[ox * stride], wv, s[ox]);
                                    }
                                }
                        }
                        if (bi == 0) {
                            if (bias_mode == 1 && b)
                                for (int ox = 0; ox < OW; ox++) tot[ox] = s[ox] + b[o];
                            else
                                memcpy(tot, s, sizeof(float) * OW);
                        } else {
                            for (int ox = 0; ox < OW; ox++) tot[ox] = tot[ox] + s[ox];
                        }
                    }
                    float* yr = y + (((size_t)n * O + o) * OH + oy) * OW;
                    if (bias_mode == 0 && b)
                        for (int ox = 0; ox < OW; ox++) yr[ox] = tot[ox] + b[o];
                    else
                        memcpy(yr, tot, sizeof(float) * OW);
                }
        free(tot);
    }
}

/* general im2col order with explicit K boundaries (in units of k = c*KH*KW + ky*KW + kx): MKL's sgemm blocks need not fall on
 * channel boundaries.  kb[0..nb]; out = (S_0 + bias) + S_1 + ... */
void orc_conv_im2col_kblocks(const float* x, int N, int C, int H, int W, const float* w, int O, int KH, int KW, const float* b,
                             int stride, int pad, const int* kb, int nb, float* y, int OH, int OW)
{
#pragma omp parallel for collapse(3) schedule(static)
    for (int n = 0; n < N; n++)
        for (int o = 0; o < O; o++)
            for (int oy = 0; oy < OH; oy++)
                for (int ox = 0; ox < OW; ox++) {
                    float tot = 0.f;
                    for (int bi = 0; bi < nb; bi++) {
                        float s = 0.f;
                        for (int k = kb[bi]; k < kb[bi + 1]; k++) {
                            const int c = k / (KH * KW), ky = (k / KW) % KH, kx = k % KW;
                            const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                            s = fmaf(x[(((size_t)n * C + c) * H + iy) * W + ix], w[(size_t)o * C * KH * KW + k], s);
                        }
                        tot = bi == 0 ? (b ? s + b[o] : s) : tot + s;
                    }
                    y[(((size_t)n * O + o) * OH + oy) * OW + ox] = tot;
                }
}

/* conv_transpose2d, stride 1: w [C][O][KH][KW]; blocks of 16 input channels, ky -> kx -> c ascending, bias at the very end */
void orc_deconv_s1(const float* x, int N, int C, int H, int W, const float* w, int O, int KH, int KW, const float* b, int pad,
                   float* y)
{
    const int OH = H - 1 - 2 * pad + KH, OW = W - 1 - 2 * pad + KW;
#pragma omp parallel
    {
        float* tot = (float*)malloc(sizeof(float) * (size_t)OW * 2);
        float* s = tot + OW;
#pragma omp for collapse(3) schedule(static)
        for (int n = 0; n < N; n++)
            for (int o = 0; o < O; o++)
                for (int oy = 0; oy < OH; oy++) {
                    int first = 1;
                    for (int cb = 0; cb < C; cb += 16) {
                        for (int ox = 0; ox < OW; ox++) s[ox] = 0.f;
                        for (int ky = 0; ky < KH; ky++) {
                            const int iy = oy + pad - ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < KW; kx++) {
                                int lo = 0, hi = OW;
                                while (lo < OW && lo + pad - kx < 0) lo++;
                                while (hi > lo && (hi - 1) + pad - kx >= W) hi--;
                                for (int c = cb; c < cb + 16 && c < C; c++) {
                                    const float wv = w[(((size_t)c * O + o) * KH + ky) * KW + kx];
                                    const float* xr = x + (((size_t)n * C + c) * H + iy) * W + pad - kx;
                                    for (int ox = lo; ox < hi; ox++) s[ox] = fmaf(xr[ox], wv, s[ox]);
                                }
                            }
                        }
                        if (first) {
                            memcpy(tot, s, sizeof(float) * OW);
                            first = 0;
                        } else {
                            for (int ox = 0; ox < OW; ox++) tot[ox] = tot[ox] + s[ox];
                        }
                    }
                    float* yr = y + (((size_t)n * O + o) * OH + oy) * OW;
                    if (b)
                        for (int ox = 0; ox < OW; ox++) yr[ox] = tot[ox] + b[o];
                    else
                        memcpy(yr, tot, sizeof(float) * OW);
                }
        free(tot);
    }
}

/* exp(x) as Sleef's expf_u10 (the vector exp of at::vec::Vectorized<float>, which torch.sigmoid's CPU kernel calls) */
static inline float pow2if(int q)
{
    union {
        int32_t i;
        float f;
    } u;
    u.i = (q + 0x7f) << 23;
    return u.f;
}
float orc_expf_u10(float d)
{
    const int q = (int)rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    float s = fmaf((float)q, -0.693145751953125f, d);
    s = fmaf((float)q, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    u = u * pow2if(q >> 1) * pow2if(q - (q >> 1));
    if (d < -104.f) u = 0.f;
    if (d > 100.f) u = INFINITY;
    return u;
}
/* torch.sigmoid, vector body: 1 / (1 + exp(0 - x)) */
void orc_sigmoid(const float* x, float* y, int64_t n)
{
    for (int64_t i = 0; i < n; i++) y[i] = 1.0f / (1.0f + orc_expf_u10(0.0f - x[i]));
}

/* expf as the C library computes it (glibc >= 2.27, sysdeps/ieee754/flt-32/e_expf.c = the published ARM optimized-routines
 * algorithm): x * 32 / ln 2 = k + r, exp(x) = 2^(k/32) * p(r) in double, one rounding to float at the end.  The 32-entry table
 * is bits(2^(i/32)) - (i << 47).  Checked against this container's libm on 4 x 10^8 random arguments in [-30, 30]: no
 * difference (tests/test_oracle_arith.py checks a sample).  Behind torch.sigmoid's SCALAR path, see orc_sigmoid_tensor. */
static const uint64_t orc_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
float orc_expf_libm(float x)
{
    if (x != x) return x;
    if (x > 88.72283f) return INFINITY;
    if (x < -103.97208f) return 0.0f;
    const double z0 = (0x1.71547652b82fep+0 * 32) * (double)x;
    double kd = z0 + 0x1.8p+52;
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= 0x1.8p+52;
    const double r = z0 - kd;
    uint64_t t = orc_exp2f_tab[ki % 32] + (ki << 47);
    double sc;
    memcpy(&sc, &t, 8);
    const double z = (0x1.c6af84b912394p-5 / 32 / 32 / 32) * r + (0x1.ebfce50fac4f3p-3 / 32 / 32);
    const double r2 = r * r;
    double y = (0x1.62e42ff0c52d6p-1 / 32) * r + 1.0;
    y = z * r2 + y;
    return (float)(y * sc);
}
/* Is element i of a contiguous tensor of n elements handled by the SCALAR tail of an ATen vectorised elementwise loop?
 * TensorIterator::for_each runs serially below 32768 elements (or with one thread); otherwise at::parallel_for gives
 * min(threads, ceil(n / 32768)) tasks ceil(n / tasks) elements each (ParallelOpenMP.h).  Inside a task's range
 * vectorized_loop (cpu/Loops.h) takes two 16-lane vectors per step and hands the last len % 32 elements to the scalar op. */
int orc_aten_scalar_tail(int64_t i, int64_t n, int threads)
{
    int64_t tasks = 1;
    if (n >= 32768 && threads > 1) {
        tasks = (n + 32767) / 32768;
        if (tasks > threads) tasks = threads;
    }
    const int64_t chunk = (n + tasks - 1) / tasks, c0 = (i / chunk) * chunk;
    const int64_t len = (n - c0 < chunk) ? n - c0 : chunk;
    return (i - c0) >= len - (len % 32);
}
/* torch.sigmoid over a whole contiguous tensor, as `threads` CPU threads compute it: Sleef's vector exp everywhere except in the
 * scalar tails, where it is 1 / (1 + expf(-x)) with the C library's expf (UnaryOpsKernel.cpp sigmoid_kernel: the scalar lambda) */
void orc_sigmoid_tensor(const float* x, float* y, int64_t n, int threads)
{
    for (int64_t i = 0; i < n; i++)
        y[i] = orc_aten_scalar_tail(i, n, threads) ? 1.0f / (1.0f + orc_expf_libm(-x[i])) : 1.0f / (1.0f + orc_expf_u10(0.0f - x[i]));
}

/* ---- torch's CPU bilinear upsampling (UpSampleKernel.cpp), align_corners = False -------------------------------------------
 * x [C][h][w] -> y [C][H][W].  The source index is one fused multiply-add; the taps are combined in one of two ways:
 *   H + W > 128  (generic separable kernel):      t_k = fma(v_k0, lx0, v_k1 * lx1);  out = fma(t_0, ly0, t_1 * ly1)
 *   H + W <= 128 (channels-last vector kernel):   w_ij = ly_i * lx_j;
 *        channels < C - C % 16:  fma(w00, v00, fma(w01, v01, fma(w11, v11, w10 * v10)))
 *        the C % 16 tail:        fma(w11, v11, fma(w10, v10, fma(w00, v00, w01 * v01)))                                  */
void orc_bilinear(const float* x, int C, int h, int w, float* y, int H, int W)
{
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (int c = 0; c < C; c++)
        for (int oy = 0; oy < H; oy++)
            for (int ox = 0; ox < W; ox++) {
                float fy = fmaf(sy, (float)oy + 0.5f, -0.5f), fx = fmaf(sx, (float)ox + 0.5f, -0.5f);
                if (fy < 0) fy = 0;
                if (fx < 0) fx = 0;
                int y0 = (int)fy, x0 = (int)fx;
                if (y0 > h - 1) y0 = h - 1;
                if (x0 > w - 1) x0 = w - 1;
                const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
                float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
                if (ly1 > 1) ly1 = 1;
                if (lx1 > 1) lx1 = 1;
                if (ly1 < 0) ly1 = 0;
                if (lx1 < 0) lx1 = 0;
                const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
                const float* b = x + (size_t)c * h * w;
                const float v00 = b[y0 * w + x0], v01 = b[y0 * w + x1], v10 = b[y1 * w + x0], v11 = b[y1 * w + x1];
                float o;
                if (H + W > 128) {
                    const float t0 = fmaf(v00, lx0, v01 * lx1), t1 = fmaf(v10, lx0, v11 * lx1);
                    o = fmaf(t0, ly0, t1 * ly1);
                } else {
                    const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
                    if (c < C - C % 16) o = fmaf(w00, v00, fmaf(w01, v01, fmaf(w11, v11, w10 * v10)));
                    else o = fmaf(w11, v11, fmaf(w10, v10, fmaf(w00, v00, w01 * v01)));
                }
                y[((size_t)c * H + oy) * W + ox] = o;
            }
}

/* ---- mean over a contiguous row: ATen's cascade sum (SumKernel.cpp: vectorized_inner_sum / row_sum / multi_row_sum), 8-lane
 * vectors (4 interleaved accumulator vectors, cascade levels of 16 steps), rows shorter than 8 in its scalar form; then / n  */
static int orc_ceil_log2(long x)
{
    if (x <= 2) return 1;
    int r = 0;
    long v = x - 1;
    while (v > 0) {
        v >>= 1;
        r++;
    }
    return r;
}
float orc_mean_row(const float* x, int n)
{
    const int V = n < 8 ? 1 : 8;
    const int vec_size = n / V, size_ilp = vec_size / 4;
    float part[4][8];
    memset(part, 0, sizeof(part));
    {
        int lp = orc_ceil_log2(size_ilp) / 4;
        if (lp < 4) lp = 4;
        const int step = 1 << lp;
        const long mask = step - 1;
        float acc[4][4][8];
        memset(acc, 0, sizeof(acc));
        int i = 0;
        for (; i + step <= size_ilp;) {
            for (int j = 0; j < step; j++, i++)
                for (int k = 0; k < 4; k++)
                    for (int l = 0; l < V; l++) acc[0][k][l] += x[((long)i * 4 + k) * V + l];
            for (int j = 1; j < 4; j++) {
                for (int k = 0; k < 4; k++)
                    for (int l = 0; l < V; l++) {
                        acc[j][k][l] += acc[j - 1][k][l];
                        acc[j - 1][k][l] = 0;
                    }
                if ((i & (mask << (j * lp))) != 0) break;
            }
        }
        for (; i < size_ilp; i++)
            for (int k = 0; k < 4; k++)
                for (int l = 0; l < V; l++) acc[0][k][l] += x[((long)i * 4 + k) * V + l];
        for (int j = 1; j < 4; j++)
            for (int k = 0; k < 4; k++)
                for (int l = 0; l < V; l++) acc[0][k][l] += acc[j][k][l];
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < V; l++) part[k][l] = acc[0][k][l];
    }
    for (int v = size_ilp * 4; v < vec_size; v++)
        for (int l = 0; l < V; l++) part[0][l] += x[(long)v * V + l];
    for (int k = 1; k < 4; k++)
        for (int l = 0; l < V; l++) part[0][l] += part[k][l];
    float fin = 0.f;
    for (int k = vec_size * V; k < n; k++) fin += x[k];
    for (int l = 0; l < V; l++) fin += part[0][l];
    return fin / (float)n;
}
void orc_mean_rows(const float* x, int rows, int n, float* y)
{
    for (int r = 0; r < rows; r++) y[r] = orc_mean_row(x + (size_t)r * n, n);
}

/* ---- y = W x for ONE input vector (nn.Linear with batch 1 -> MKL sgemm with n = 1): each output row is a dot product in one
 * of three orders, depending on where the row falls in MKL's row partition (class per row: measured, refarith_tables.json):
 *   0 "main":  lane l of ONE 16-lane accumulator takes elements 1 + 16 v + l; element 0 starts lane 0's chain; lanes reduced
 *              l + 8, l + 4, l + 2, l + 1; then the (K - 1) % 16 tail elements as a second vector whose lane 0 continues from
 *              the body's sum, reduced the same way
 *   2 "rem2":  two accumulators (elements 1 + 32 j + l and 17 + 32 j + l), added lane-wise, leftover vectors / tail fma'd in,
 *              lanes reduced, element 0's product added last
 *   1 "rem1":  the same with one accumulator                                                                                */
static float orc_hred16(float* a)
{
    for (int s = 8; s >= 1; s /= 2)
        for (int l = 0; l < s; l++) a[l] = a[l] + a[l + s];
    return a[0];
}
float orc_dot_main(const float* w, const float* x, int K)
{
    float acc[16];
    memset(acc, 0, sizeof(acc));
    acc[0] = w[0] * x[0];
    const int nb = (K - 1) / 16, nt = (K - 1) % 16;
    for (int v = 0; v < nb; v++)
        for (int l = 0; l < 16; l++) {
            const int k = 1 + v * 16 + l;
            acc[l] = fmaf(w[k], x[k], acc[l]);
        }
    float S = orc_hred16(acc);
    if (nt) {
        float t[16];
        memset(t, 0, sizeof(t));
        for (int l = 0; l < nt; l++) {
            const int k = 1 + nb * 16 + l;
            t[l] = l == 0 ? fmaf(w[k], x[k], S) : w[k] * x[k];
        }
        S = orc_hred16(t);
    }
    return S;
}
float orc_dot_rem(const float* w, const float* x, int K, int U)
{
    float A[2][16];
    memset(A, 0, sizeof(A));
    const int step = U * 16, n2 = (K - 1) / step;
    for (int j = 0; j < n2; j++)
        for (int u = 0; u < U; u++)
            for (int l = 0; l < 16; l++) {
                const int k = 1 + step * j + u * 16 + l;
                A[u][l] = fmaf(w[k], x[k], A[u][l]);
            }
    int pos = 1 + step * n2, rem = K - pos;
    float acc[16];
    for (int l = 0; l < 16; l++) acc[l] = U == 2 ? A[0][l] + A[1][l] : A[0][l];
    while (rem >= 16) {
        for (int l = 0; l < 16; l++) acc[l] = fmaf(w[pos + l], x[pos + l], acc[l]);
        pos += 16;
        rem -= 16;
    }
    for (int l = 0; l < rem; l++) acc[l] = fmaf(w[pos + l], x[pos + l], acc[l]);
    const float S = orc_hred16(acc);
    return S + w[0] * x[0];
}
void orc_linear_b1(const float* W, const float* x, int J, int K, const int* row_class, float* y)
{
    for (int j = 0; j < J; j++) {
        const int c = row_class ? row_class[j] : 0;
        y[j] = c == 0 ? orc_dot_main(W + (size_t)j * K, x, K) : orc_dot_rem(W + (size_t)j * K, x, K, c);
    }
}

/* ---- y = x W^T for a batch of TWO vectors (nn.Linear, bias-free: MKL sgemm with n = 2; measured like the batch-1 classes: lane
 * membership with the absorbing probe, the reduction tree with +BIG / -BIG / 1 triples): every output, both batch rows alike --
 *   K < 48:  one k-ordered fma chain from 0;
 *   else:    lane l of ONE 16-lane accumulator takes elements 16 v + l, the K % 16 tail as one more (masked) fma step; then
 *            q_i = ((a_i + a_{i+4}) + a_{i+8}) + a_{i+12} for i = 0..3, and (q_0 + q_1) + (q_2 + q_3).
 * Batches of 3 and more follow other forms (not measured: no reference golden has them). */
float orc_dot_b2(const float* w, const float* x, int K)
{
    if (K < 48) {
        float s = 0.f;
        for (int k = 0; k < K; k++) s = fmaf(w[k], x[k], s);
        return s;
    }
    float a[16];
    memset(a, 0, sizeof(a));
    const int nb = K / 16;
    for (int v = 0; v < nb; v++)
        for (int l = 0; l < 16; l++) a[l] = fmaf(w[16 * v + l], x[16 * v + l], a[l]);
    for (int l = 0; l < K - 16 * nb; l++) a[l] = fmaf(w[16 * nb + l], x[16 * nb + l], a[l]);
    float q[4];
    for (int i = 0; i < 4; i++) q[i] = ((a[i] + a[i + 4]) + a[i + 8]) + a[i + 12];
    return (q[0] + q[1]) + (q[2] + q[3]);
}
void orc_linear_b2(const float* W, const float* x, int J, int K, float* y)
{
    for (int b = 0; b < 2; b++)
        for (int j = 0; j < J; j++) y[(size_t)b * J + j] = orc_dot_b2(W + (size_t)j * K, x + (size_t)b * K, K);
}

/* ---- conv_transpose2d, stride 2, kernel k, padding k/2, output_padding 1 (oneDNN brg_deconv + brgconv_strided) ----------------
 * w [C][O][k][k].  Output pixel (oy, ox) of phase (py, px) = (oy & 1, ox & 1) gathers the taps with (oy + pad - ky) even and
 * (ox + pad - kx) even.  The taps are accumulated in CHAINS: inside a chain every tap continues the fma chain of the previous
 * one, channels ascending inside a tap; a new chain starts from 0; the chain sums are added in order; the bias comes last.
 * Which taps share a chain depends on the layer shape and on the column block of the output pixel (measured: a "recipe" per
 * (py, px, j = ox / 2)): desc[desc_off[(py * 2 + px) * Win + j]] = n, then n x {ky, kx, fresh}.  Taps outside the image add
 * nothing (exact zeros). */
void orc_deconv_s2(const float* x, int N, int C, int H, int W, const float* w, int O, int k, const float* b, const int* desc_off,
                   const int* desc, float* y)
{
    const int pad = k / 2, OH = 2 * H, OW = 2 * W;
#pragma omp parallel for collapse(3) schedule(static)
    for (int n = 0; n < N; n++)
        for (int o = 0; o < O; o++)
            for (int oy = 0; oy < OH; oy++)
                for (int ox = 0; ox < OW; ox++) {
                    const int* d = desc + desc_off[((oy & 1) * 2 + (ox & 1)) * W + ox / 2];
                    const int nt = d[0];
                    float tot = 0.f, s = 0.f;
                    int have = 0;
                    for (int t = 0; t < nt; t++) {
                        const int ky = d[1 + 3 * t], kx = d[2 + 3 * t], fresh = d[3 + 3 * t];
                        if (fresh && t > 0) {
                            tot = have ? tot + s : s;
                            have = 1;
                            s = 0.f;
                        }
                        const int ty = oy + pad - ky, tx = ox + pad - kx;
                        if (ty < 0 || tx < 0 || (ty & 1) || (tx & 1)) continue;
                        const int iy = ty / 2, ix = tx / 2;
                        if (iy >= H || ix >= W) continue;
                        for (int c = 0; c < C; c++)
                            s = fmaf(x[(((size_t)n * C + c) * H + iy) * W + ix], w[(((size_t)c * O + o) * k + ky) * k + kx], s);
                    }
                    tot = have ? tot + s : s;
                    y[(((size_t)n * O + o) * OH + oy) * OW + ox] = b ? tot + b[o] : tot;
                }
}
