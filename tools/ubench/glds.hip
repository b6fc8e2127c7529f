// Semantics probe for __builtin_amdgcn_global_load_lds on gfx950: per-lane source, wave-uniform LDS base + lane*16,
// behaviour of EXEC-masked lanes, visibility after __syncthreads().
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* __restrict__ g, float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * 256 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    for (int i = tid; i < 2 * 256 * 4; i += 256) lds[i] = -1.f;
    __syncthreads();
    for (int u = 0; u < 2; ++u) {
        const int f = tid + u * 256;             // float4 slot in LDS
        const int src = (f * 7 + 3) % 512;       // permuted source slot
        float* dst = lds + (wave * 64 + u * 256) * 4;  // wave-uniform base; lane i lands at base + i*16 bytes
        if ((lane % 5) != 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + src * 4),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        else
            *reinterpret_cast<f32x4*>(lds + f * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    for (int i = tid; i < 2 * 256 * 4; i += 256) out[i] = lds[i];
}
int main()
{
    float *g, *o; static float hg[512 * 4], ho[512 * 4];
    for (int i = 0; i < 512 * 4; ++i) hg[i] = (float)i;
    hipMalloc(&g, sizeof(hg)); hipMalloc(&o, sizeof(ho)); hipMemcpy(g, hg, sizeof(hg), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, g, o); hipDeviceSynchronize();
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int f = 0; f < 512; ++f) {
        const int lane = f & 63; const int src = (f * 7 + 3) % 512;
        for (int e = 0; e < 4; ++e) {
            const float want = (lane % 5) ? hg[src * 4 + e] : 0.f;
            if (ho[f * 4 + e] != want) { if (bad < 8) printf("slot %d e %d got %g want %g\n", f, e, ho[f * 4 + e], want); ++bad; }
        }
    }
    printf("glds probe: %d mismatches\n", bad);
    return bad != 0;
}
