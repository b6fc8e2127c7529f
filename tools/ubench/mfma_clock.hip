// Sustained shader clock and fp32 MFMA rate under a long matrix-core load with live operand data (the 157.3 TF/s peak
// assumes 2.4 GHz): every wave loops over v_mfma_f32_16x16x4_f32 with per-lane pseudo-random operands that change every
// iteration, and brackets the loop with the shader cycle counter (clock64) and the 100 MHz real-time counter
// (wall_clock64).  Prints effective MHz and TFLOP/s for a short and a long burn at 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 mfma_clock.hip -o mfma_clock && ./mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void burn(float* out, unsigned long long* clk, int iters, unsigned seed)
{
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned r = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        r = r * 1664525u + 1013904223u;
        a[i] = (float)(int)(r >> 8) * (1.0f / 8388608.0f) - 1.0f;
        r = r * 1664525u + 1013904223u;
        b[i] = (float)(int)(r >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i + it) & 3], acc[i], 0, 0, 0);
        a[it & 3] = -a[it & 3] * 0.999f;  // operands keep toggling; magnitudes stay bounded
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}
int main()
{
    float* d; (void)hipMalloc(&d, 64);
    unsigned long long* dc; (void)hipMalloc(&dc, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2}) {
        for (int iters : {20000, 400000}) {
            const int blocks = 256 * wps;
            hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, d, dc, 1000, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, d, dc, iters, 7u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
            const double flops = (double)blocks * 4 * iters * 12 * 2048.0;
            printf("%d wave(s)/SIMD, %6d iterations: %8.2f ms, %6.1f TFLOP/s, shader clock %.0f MHz (cycles / 100 MHz real time)\n", wps,
                   iters, ms, flops / ms / 1e9, (double)h[0] / ((double)h[1] / 100.0));
        }
    }
    return 0;
}
