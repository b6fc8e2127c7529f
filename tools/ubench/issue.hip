// Single-wave issue-rate probes on gfx950: independent vs dependent SALU ops, s_mul latency, v_mad_u64_u32 latency.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
__global__ void probe(unsigned long long* out, int seed)
{
    unsigned long long t0, t1;
    unsigned a = __builtin_amdgcn_readfirstlane(seed), b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
    // 1. eight independent SALU adds per group
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 128; ++it) {
        asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n"
                     "s_add_u32 %4, %4, 3\n s_add_u32 %5, %5, 3\n s_add_u32 %6, %6, 3\n s_add_u32 %7, %7, 3\n"
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h)::"scc");
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    // 2. dependent s_mul_i32 chain
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 128; ++it) { REP8(asm volatile("s_mul_i32 %0, %0, %1" : "+s"(a) : "s"(b));) }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[1] = t1 - t0;
    // 3. dependent s_mul_hi_u32 chain
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 128; ++it) { REP8(asm volatile("s_mul_hi_u32 %0, %0, %1\n s_or_b32 %0, %0, 0x10000000" : "+s"(a) : "s"(b) : "scc");) }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[2] = t1 - t0;
    // 4. eight independent s_mul_i32
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 128; ++it) {
        asm volatile("s_mul_i32 %0, %0, %0\n s_mul_i32 %1, %1, %1\n s_mul_i32 %2, %2, %2\n s_mul_i32 %3, %3, %3\n"
                     "s_mul_i32 %4, %4, %4\n s_mul_i32 %5, %5, %5\n s_mul_i32 %6, %6, %6\n s_mul_i32 %7, %7, %7\n"
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h));
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[3] = t1 - t0;
    // 5. dependent v_mad_u64_u32 chain
    unsigned long long v = threadIdx.x + seed;
    unsigned m = threadIdx.x * 3 + 1;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 128; ++it) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(v) : "v"(m) : "vcc");) }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[4] = t1 - t0;
    // 6. eight independent VALU adds
    unsigned v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 128; ++it) {
        asm volatile("v_add_u32 %0, %0, 3\n v_add_u32 %1, %1, 3\n v_add_u32 %2, %2, 3\n v_add_u32 %3, %3, 3\n"
                     "v_add_u32 %4, %4, 3\n v_add_u32 %5, %5, 3\n v_add_u32 %6, %6, 3\n v_add_u32 %7, %7, 3\n"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[5] = t1 - t0;
    // 7. alternating independent SALU / VALU
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 128; ++it) {
        asm volatile("s_add_u32 %0, %0, 3\n v_add_u32 %4, %4, 3\n s_add_u32 %1, %1, 3\n v_add_u32 %5, %5, 3\n"
                     "s_add_u32 %2, %2, 3\n v_add_u32 %6, %6, 3\n s_add_u32 %3, %3, 3\n v_add_u32 %7, %7, 3\n"
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"scc");
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[6] = t1 - t0; out[7] = a + b + c + d + e + f + g + h + v + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7; }
    // 8. v_readlane with SGPR consumed by a VALU op, result feeding the next readlane's source
    unsigned vv = threadIdx.x, ss;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 128; ++it) { REP8(asm volatile("v_readlane_b32 %1, %0, 5\n v_add_u32 %0, %1, %0" : "+v"(vv), "=&s"(ss));) }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[8] = t1 - t0; out[7] += vv; }
}
int main()
{
    unsigned long long* d; (void)hipMalloc(&d, 128); unsigned long long h[16];
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 5 + r); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    const char* names[] = {"8 indep s_add", "dep s_mul_i32", "dep s_mul_hi+or (2 instr)", "8 indep s_mul_i32", "dep v_mad_u64_u32",
                           "8 indep v_add", "alternating s/v indep", "", "readlane->v_add round (2 instr)"};
    for (int i : {0, 1, 2, 3, 4, 5, 6, 8}) printf("%-34s %8llu ticks / 1024 = %.2f per instr(or round)\n", names[i], h[i], h[i] / 1024.0);
    return 0;
}
