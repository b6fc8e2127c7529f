// Dependent-chain latency of the instructions a single-wave serial coder loop is made of, on gfx950.
// One wavefront, long dependent chains, bracketed by s_memrealtime (100 MHz): ns per instruction (or per group).
//   hipcc --offload-arch=gfx950 -O2 oplat.hip -o oplat && ./oplat
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define N_OUT 48
#define ITERS 256  // x64 = 16384 groups per probe

#define PROBE(slot, body)                                                        \
    {                                                                            \
        unsigned long long w0 = __builtin_amdgcn_s_memrealtime();                \
        for (int it = 0; it < ITERS; ++it) { REP64(body) }                       \
        asm volatile("s_nop 0" ::"s"(s), "v"(v), "v"(v64), "s"(s64));            \
        unsigned long long w1 = __builtin_amdgcn_s_memrealtime();                \
        if (threadIdx.x == 0) out[slot] = w1 - w0;                               \
    }

__global__ __launch_bounds__(64) void probe(unsigned long long* out, int seed)
{
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = ((i * 7 + 1) & 1023) * 4;
    __syncthreads();
    unsigned s = __builtin_amdgcn_readfirstlane(seed) | 1u, s2 = 12345u | s;
    unsigned v = threadIdx.x + seed, v2 = threadIdx.x * 3 + 7;
    unsigned long long v64 = ((unsigned long long)(seed + 1) << 33) | threadIdx.x;
    unsigned long long s64 = ((unsigned long long)s << 35) | 99u;
    double d = 1.0 + seed * 1e-9, d2 = 1.0000001;
    float f = 1.0f + seed * 1e-6f;
    // ---- VALU dependent chains
    PROBE(0, asm volatile("v_add_u32 %0, %0, 3" : "+v"(v));)
    PROBE(1, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v64) : "v"(v2), "s"(s) : "vcc");)
    PROBE(2, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v) : "v"(v2));)
    PROBE(3, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v) : "v"(v2));)
    PROBE(4, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v) : "v"(v2));)
    PROBE(5, asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(v) : "v"(v2));)
    PROBE(6, asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshlrev_b64 %0, 1, %0" : "+v"(v64));)  // two 64-bit shifts
    PROBE(7, asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, 0, vcc" : "+v"(v), "+v"(v2) : "v"(v2) : "vcc");)
    PROBE(8, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(d2));)
    PROBE(9, asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f));)
    PROBE(10, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(v2) : "vcc");)
    PROBE(11, asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(v) : "v"(v2));)
    // ---- SALU dependent chains
    PROBE(12, asm volatile("s_add_u32 %0, %0, 3" : "+s"(s)::"scc");)
    PROBE(13, asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s) : "s"(s2));)
    PROBE(14, asm volatile("s_mul_hi_u32 %0, %0, %1" : "+s"(s) : "s"(s2));)
    PROBE(15, asm volatile("s_lshr_b64 %0, %0, 1\n s_lshl_b64 %0, %0, 1" : "+s"(s64)::"scc");)
    PROBE(16, asm volatile("s_cmp_lt_u32 %0, %1\n s_cselect_b32 %0, %1, %0" : "+s"(s) : "s"(s2) : "scc");)
    PROBE(17, asm volatile("s_bfe_u32 %0, %0, 0x100001" : "+s"(s)::"scc");)
    // ---- crossings
    PROBE(18, asm volatile("v_readfirstlane_b32 %0, %1\n v_mov_b32 %1, %0" : "+s"(s), "+v"(v));)             // V->S->V
    PROBE(19, asm volatile("s_and_b32 %0, %0, 63\n v_readlane_b32 %0, %1, %0" : "+s"(s) : "v"(v2) : "scc");)  // readlane with SGPR select
    PROBE(20, asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_bcnt1_i32_b64 %0, vcc" : "+s"(s) : "v"(v2) : "vcc", "scc");)  // S->cmp->bcnt
    PROBE(21, asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_cbranch_vccz 1f\n s_add_u32 %0, %0, 1\n1:\n" : "+s"(s) : "v"(v2) : "vcc", "scc");)
    PROBE(22, asm volatile("v_writelane_b32 %0, %1, 3\n v_readlane_b32 %1, %0, 3" : "+v"(v), "+s"(s));)
    PROBE(23, asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 2\n1:\n" : "+s"(s)::"scc");)  // not-taken branch
    PROBE(24, asm volatile("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 2\n1:\n s_or_b32 %0, %0, 1" : "+s"(s)::"scc");)  // taken fwd branch
    // ---- LDS
    {
        unsigned idx = (s & 1023) * 4;
        unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < ITERS * 16; ++it)
            asm volatile("v_mov_b32 %1, %0\n ds_read_b32 %1, %1\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %0, %1" : "+s"(idx), "+v"(v));
        unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) out[25] = (w1 - w0) * 4;  // normalised to ITERS*64 groups
        s += idx;
    }
    // ---- throughput (independent) of the quarter-rate candidates: 4 independent chains
    {
        unsigned long long a = v64, b = v64 + 1, c = v64 + 2, e = v64 + 3;
        unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < ITERS; ++it) {
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                               "v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(e) : "v"(v2), "s"(s2) : "vcc");)
        }
        unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) out[26] = w1 - w0;  // 64 mads per REP16 group -> same count as a PROBE
        v64 += a + b + c + e;
    }
    {
        unsigned a = v, b = v + 1, c = v + 2, e = v + 3;
        unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < ITERS; ++it) {
            REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(e) : "v"(v2));)
        }
        unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) out[27] = w1 - w0;
        v += a + b + c + e;
    }
    {
        unsigned a = s, b = s + 1, c = s + 2, e = s + 3;
        unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < ITERS; ++it) {
            REP16(asm volatile("s_mul_i32 %0, %0, %4\n s_mul_hi_u32 %1, %1, %4\n s_mul_i32 %2, %2, %4\n s_mul_hi_u32 %3, %3, %4"
                               : "+s"(a), "+s"(b), "+s"(c), "+s"(e) : "s"(s2));)
        }
        unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) out[28] = w1 - w0;
        s += a + b + c + e;
    }
    // backward taken branch (loop edge) cost: tight loop of s_sub + s_cbranch
    {
        unsigned cnt = ITERS * 64;
        unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        asm volatile("1:\n s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b" : "+s"(cnt)::"scc");
        unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) out[29] = w1 - w0;
    }
    if (threadIdx.x == 0) out[N_OUT - 1] = s + v + (unsigned)v64 + (unsigned)s64 + (unsigned)d + (unsigned)f + v2;
}

int main()
{
    unsigned long long* dv;
    hipMalloc(&dv, N_OUT * 8);
    unsigned long long h[N_OUT];
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dv, 5 + r);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, dv, N_OUT * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"v_add_u32", "v_mad_u64_u32 (acc chain)", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24",
                           "v_lshr_b64 + v_lshl_b64 (pair)", "v_add_co + v_addc (pair)", "v_fma_f64", "v_fma_f32",
                           "v_cmp + v_cndmask (pair)", "v_mul_hi_u32_u24", "s_add_u32", "s_mul_i32", "s_mul_hi_u32",
                           "s_lshr_b64 + s_lshl_b64 (pair)", "s_cmp + s_cselect (pair)", "s_bfe_u32",
                           "v_readfirstlane + v_mov (V->S->V)", "s_and + v_readlane(sgpr sel)", "v_cmp(sgpr) + s_bcnt1", "v_cmp + s_cbranch_vccz(+add)",
                           "v_writelane + v_readlane", "s_cmp + branch not taken + add", "s_cmp + branch taken + or", "v_mov+ds_read+wait+readfirstlane",
                           "4 indep v_mad_u64_u32 (per instr)", "4 indep v_mul_hi_u32 (per instr)", "4 indep s_mul (per instr)",
                           "loop edge: s_sub+s_cmp+s_cbranch taken"};
    const double n = ITERS * 64.0;
    for (int i = 0; i < 30; ++i) printf("%-44s %7.2f ns  (%5.1f clk @2.4GHz)\n", names[i], h[i] * 10.0 / n, h[i] * 10.0 / n * 2.4);
    return 0;
}
