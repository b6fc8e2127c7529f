// What would emulating the convolutions' fp32 MFMA with split bf16 operands buy, and what would it cost in accuracy?
// (Both reviews list it as an optional experiment; the product path stays on v_mfma_f32_16x16x4_f32: a different rounding
// of every sum re-rolls every near-boundary symbol of the parity goldens.)
//   x = h + m + l, each part a bf16 (round to nearest even of what is left): 24 mantissa bits in three pieces.
//   x3: hh + hm + mh                      (terms down to 2^-16 relative dropped)
//   x6: hh + hm + mh + hl + lh + mm       (terms of 2^-24 and below dropped)
// Part 1 (accuracy): one 64 x 64 tile over K = 1152 (a 3x3 layer on 128 channels) of N(0,1) data, against a double
// reference: fp32 MFMA chain (v_mfma_f32_32x32x2_f32), x3, x6 -- max and rms error relative to sum |a||b|.
// Part 2 (speed): the main-loop shape of the conv kernel with fragments from LDS (wave tile 64 x 64 = 2 x 2 MFMA tiles of
// 32 x 32), fp32 (32 v_mfma_f32_32x32x2_f32 per 16 channels) against x6 (24 v_mfma_f32_32x32x16_bf16) and x3 (12), 512
// workgroups of 4 waves: fp32-equivalent TFLOP/s.
//   hipcc --offload-arch=gfx950 -O2 bf16_split.hip -o bf16_split && ./bf16_split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __host__ inline float bf16_rne(float x)  // x rounded to bf16, as a float
{
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    u &= 0xFFFF0000u;
    float r;
    memcpy(&r, &u, 4);
    return r;
}
__device__ inline __bf16 to_bf16(float x_already_bf16)
{
    unsigned u;
    memcpy(&u, &x_already_bf16, 4);
    unsigned short h = (unsigned short)(u >> 16);
    __bf16 r;
    memcpy(&r, &h, 2);
    return r;
}

// ---- part 1: one wave, one 32 x 32 tile per (ti, tj); A [64][K], B [64][K] row-major ---------------------------------------
__global__ void tile_f32(const float* A, const float* B, int K, float* C)
{
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5, ti = blockIdx.x >> 1, tj = blockIdx.x & 1;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(ti * 32 + l31) * K + k + h], B[(tj * 32 + l31) * K + k + h], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[(ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + tj * 32 + l31] = acc[r];
}
template <int TERMS>
__global__ void tile_split(const float* A, const float* B, int K, float* C)
{
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5, ti = blockIdx.x >> 1, tj = blockIdx.x & 1;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; k += 16) {
        bf16x8 a[3], b[3];
        for (int e = 0; e < 8; ++e) {  // lane holds k + 8 h + e of its row
            float x = A[(ti * 32 + l31) * K + k + 8 * h + e], y = B[(tj * 32 + l31) * K + k + 8 * h + e];
            for (int p = 0; p < 3; ++p) {
                const float xp = bf16_rne(x), yp = bf16_rne(y);
                a[p][e] = to_bf16(xp);
                b[p][e] = to_bf16(yp);
                x -= xp;
                y -= yp;
            }
        }
        // smallest terms first
        if (TERMS == 6) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C[(ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + tj * 32 + l31] = acc[r];
}

// ---- part 2: timing loops, fragments from LDS ----------------------------------------------------------------------------
template <int MODE>  // 0: fp32 32x32x2, 3 / 6: split bf16
__global__ __launch_bounds__(256) void loop(float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16 * 1024; i += 256) lds[i] = 1.0f + (i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const int base = (it & 7) * 1024;
        if (MODE == 0) {
            f32x4 a0[2], a1[2], b0[2], b1[2];  // 16 channels of a 32-row tile: two halves of 8 (4 per half-wave)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a0[i] = *reinterpret_cast<const f32x4*>(lds + base + (i * 128 + lane) * 4);
                a1[i] = *reinterpret_cast<const f32x4*>(lds + base + (i * 128 + 64 + lane) * 4);
                b0[i] = *reinterpret_cast<const f32x4*>(lds + 8192 + base / 2 + (i * 128 + lane) * 4);
                b1[i] = *reinterpret_cast<const f32x4*>(lds + 8192 + base / 2 + (i * 128 + 64 + lane) * 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][e], b0[j][e], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][e], b1[j][e], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            bf16x8 a[2][3], b[2][3];  // the three planes of 16 channels: one ds_read_b128 each
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    if (MODE == 3 && p == 2) continue;
                    a[i][p] = *reinterpret_cast<const bf16x8*>(lds + base + ((i * 3 + p) * 64 + lane) * 4);
                    b[i][p] = *reinterpret_cast<const bf16x8*>(lds + 8192 + base / 2 + ((i * 3 + p) * 64 + lane) * 4);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (MODE == 6) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    if (s == 12345.f) out[0] = s;
}

template <int MODE>
static void time_loop(const char* name, float* d)
{
    auto k = loop<MODE>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    const int blocks = 512, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 72 * 1024, 0, d, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 72 * 1024, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 64 * 64 * 16 * (double)iters * 4 * blocks;  // fp32-equivalent: a 64 x 64 x 16 block per wave and step
    printf("%-28s %8.2f ms  %7.1f TFLOP/s fp32-equivalent\n", name, ms, flops / ms / 1e9);
}

int main()
{
    const int K = 1152;
    float *hA = (float*)malloc(64 * K * 4), *hB = (float*)malloc(64 * K * 4), *hC = (float*)malloc(64 * 64 * 4);
    srand(1);
    auto nrm = []() {
        double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
        return (float)(sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v));
    };
    for (int i = 0; i < 64 * K; ++i) {
        hA[i] = nrm();
        hB[i] = nrm();
    }
    float *A, *B, *C;
    (void)hipMalloc(&A, 64 * K * 4);
    (void)hipMalloc(&B, 64 * K * 4);
    (void)hipMalloc(&C, 64 * 64 * 4);
    (void)hipMemcpy(A, hA, 64 * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB, 64 * K * 4, hipMemcpyHostToDevice);
    for (int v = 0; v < 3; ++v) {
        if (v == 0) hipLaunchKernelGGL(tile_f32, dim3(4), dim3(64), 0, 0, A, B, K, C);
        if (v == 1) hipLaunchKernelGGL(tile_split<3>, dim3(4), dim3(64), 0, 0, A, B, K, C);
        if (v == 2) hipLaunchKernelGGL(tile_split<6>, dim3(4), dim3(64), 0, 0, A, B, K, C);
        (void)hipMemcpy(hC, C, 64 * 64 * 4, hipMemcpyDeviceToHost);
        double mx = 0, sq = 0;
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                double ref = 0, mag = 0;
                for (int k = 0; k < K; ++k) {
                    ref += (double)hA[i * K + k] * hB[j * K + k];
                    mag += fabs((double)hA[i * K + k] * hB[j * K + k]);
                }
                const double e = fabs(hC[i * 64 + j] - ref) / mag;
                mx = e > mx ? e : mx;
                sq += e * e;
            }
        printf("%-28s error / sum|a||b|: max %.3e  rms %.3e\n", v == 0 ? "fp32 MFMA (32x32x2)" : (v == 1 ? "bf16 x3 (hh, hm, mh)" : "bf16 x6 (+ hl, lh, mm)"), mx,
               sqrt(sq / 4096));
    }
    time_loop<0>("fp32 MFMA 32x32x2", C);
    time_loop<3>("bf16 x3, 32x32x16", C);
    time_loop<6>("bf16 x6, 32x32x16", C);
    return 0;
}
