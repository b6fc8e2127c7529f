// MFMA utilisation of the conv kernel's stage structure without its global loads: per piece, 256 threads write a
// weight slab + patch rows to LDS, barrier, then 5 taps x (7 ds_read_b128 + 48 MFMA 16x16x4 f32), barrier.
// Variants: barriers on/off, LDS reads on/off.  2 blocks per CU (72 KB dynamic LDS each).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool BARRIER, bool READS, bool WRITES>
__global__ __launch_bounds__(256) void stage(float* out, int pieces)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 w = {1.f * tid, 2.f, 3.f, 4.f};
    for (int p = 0; p < pieces; ++p) {
        if (WRITES) {
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<f32x4*>(lds + (k * 256 + tid) * 4) = w;
        }
        if (BARRIER) __syncthreads();
#pragma unroll 1
        for (int t = 0; t < 5; ++t) {
            f32x4 fa[3], fb[4];
            if (READS) {
#pragma unroll
                for (int i = 0; i < 3; ++i) fa[i] = *reinterpret_cast<const f32x4*>(lds + ((t * 3 + i) * 64 + lane) * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) fb[k] = *reinterpret_cast<const f32x4*>(lds + 4096 + ((wave * 4 + k) * 64 + lane) * 4);
            } else {
                for (int i = 0; i < 3; ++i) fa[i] = w;
                for (int k = 0; k < 4; ++k) fb[k] = w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        acc[i * 4 + k] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][q], fb[k][q], acc[i * 4 + k], 0, 0, 0);
        }
        if (BARRIER) __syncthreads();
        w[0] += 1.f;
    }
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.f) out[0] = s;
}
template <bool B, bool R, bool W>
void run(const char* name, float* d)
{
    const int blocks = 512, pieces = 2000;
    auto k = stage<B, R, W>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 72 * 1024, 0, d, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 72 * 1024, 0, d, pieces);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.1f TFLOP/s\n", name, (double)blocks * 4 * pieces * 240 * 2048.0 / ms / 1e9);
}
int main()
{
    float* d; (void)hipMalloc(&d, 64);
    run<false, false, false>("mfma only", d);
    run<false, true, false>("+ lds reads", d);
    run<true, true, false>("+ lds reads + barriers", d);
    run<true, true, true>("+ lds reads + barriers + writes", d);
    return 0;
}
