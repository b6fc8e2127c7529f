// Sustained fp32 MFMA rate of the whole chip (v_mfma_f32_16x16x4_f32, no memory traffic): the practical ceiling under
// the conv kernels.  Prints TFLOP/s for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void burn(float* out, int iters, float a0, float b0)
{
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.f) out[0] = s;
}
int main()
{
    float* d; (void)hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;  // 256 CUs x wps blocks of 4 waves = wps waves per SIMD
        const int iters = 40000;
        hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, d, 1000, 1.0f, 2.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 2.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * 12 * 2048.0;
        printf("%d wave(s)/SIMD: %.1f ms, %.1f TFLOP/s\n", wps, ms, flops / ms / 1e9);
    }
    return 0;
}
