// Dependent-chain latency probes on gfx950: SALU add, VALU add, readfirstlane round trip, LDS read + readfirstlane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void probe(unsigned long long* out, int seed)
{
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 7 + 1) & 1023;
    __syncthreads();
    unsigned long long t0, t1;
    unsigned s = __builtin_amdgcn_readfirstlane(seed);
    // 1. SALU dependent adds
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) { REP16(asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) :: "scc");) }
    asm volatile("s_nop 0" ::"s"(s));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    // 2. VALU dependent adds
    unsigned v = threadIdx.x + seed;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) { REP16(asm volatile("v_add_u32 %0, %0, 3" : "+v"(v));) }
    asm volatile("" ::"v"(v));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[1] = t1 - t0;
    // 3. VALU -> readfirstlane -> SALU -> v_mov round trip
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
        REP16(asm volatile("v_readfirstlane_b32 %0, %1\n s_add_u32 %0, %0, 1\n v_mov_b32 %1, %0" : "+s"(s), "+v"(v) :: "scc");)
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[2] = t1 - t0;
    // 4. dependent LDS read chain through readfirstlane (pointer chase)
    unsigned idx = s & 1023;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 1024; ++it) idx = __builtin_amdgcn_readfirstlane(lds[idx]);
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[3] = t1 - t0; out[4] = idx + s + v; }
    // 5. 64-bit SALU multiply chain (s_mul_i32 + s_mul_hi_u32 + adds)
    unsigned long long x = ((unsigned long long)seed << 33) | 12345u;
    unsigned f = (s & 1023) | 1;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 1024; ++it) x = (unsigned long long)f * (x >> 16) + (x & 0xffff);
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[5] = t1 - t0; out[6] = x; }
    // 6. v_readlane (SGPR lane select) -> SALU -> v_readlane chain
    unsigned sel = s & 63;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
        REP16(asm volatile("v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63" : "+s"(sel) : "v"(v) : "scc");)
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[7] = t1 - t0; out[4] += sel; }
    // 7. long SALU chain bracketed by both clocks: ticks of s_memtime vs s_memrealtime (100 MHz)
    unsigned long long r0 = __builtin_readcyclecounter();
    unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 8192; ++it) { REP16(asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) :: "scc");) }
    asm volatile("s_nop 0" ::"s"(s));
    t1 = __builtin_amdgcn_s_memtime();
    unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[8] = t1 - t0; out[9] = w1 - w0; out[10] = r1 - r0; }
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 128); unsigned long long h[16];
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 5 + r); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    printf("s_memtime ticks (100 MHz? or shader clk): salu add x1024: %llu (%.2f/instr)\n", h[0], h[0] / 1024.0);
    printf("valu add x1024: %llu (%.2f/instr)\n", h[1], h[1] / 1024.0);
    printf("rfl+sadd+vmov x1024: %llu (%.2f/round)\n", h[2], h[2] / 1024.0);
    printf("lds chase x1024: %llu (%.2f/hop)\n", h[3], h[3] / 1024.0);
    printf("u64 mul chain x1024: %llu (%.2f/iter)\n", h[5], h[5] / 1024.0);
    printf("readlane+sand x1024: %llu (%.2f/round)\n", h[7], h[7] / 1024.0);
    printf("131072 salu adds: memtime %llu, memrealtime(100MHz) %llu, cyclecounter %llu -> %.1f ns/add, memtime tick = %.2f ns\n", h[8], h[9], h[10],
           h[9] * 10.0 / 131072, h[9] * 10.0 / h[8]);
    return 0;
}
