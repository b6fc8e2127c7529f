// Does v_cmpx + v_readfirstlane pick the first lane that passes the compare, back to back, on gfx950?  (The decoder's
// rows-in-lanes search wants the symbol's slot without a scalar step in between.)  One wave; per iteration the lanes hold
// a decreasing table g[lane], s runs through thresholds; expected index = first lane with s >= g[lane].
//   hipcc --offload-arch=gfx950 -O2 cmpx_first.hip -o cmpx_first && ./cmpx_first
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
__global__ __launch_bounds__(64) void k(const unsigned* g, unsigned* out_a, unsigned* out_v, int nops)
{
    const unsigned lane = threadIdx.x;
    const unsigned v = g[lane];
    unsigned res_a = 0, res_v = 0;
    for (int it = 0; it < N; ++it) {
        unsigned s = __builtin_amdgcn_readfirstlane((it * 37u) & 0xFFFFu), a, t;
#define CASE(n, NOP)                                                                                                        \
    case n:                                                                                                                 \
        asm volatile("v_cmpx_ge_u16 vcc, %2, %3\n" NOP "v_readfirstlane_b32 %0, %3\n v_readfirstlane_b32 %1, %4\n s_mov_b64 exec, -1" \
                     : "=&s"(t), "=&s"(a) : "s"(s), "v"(v), "v"(lane) : "vcc", "s84", "s85", "s86", "s87", "v57", "v58", "v59"); \
        break;
        switch (nops) {
            CASE(0, "")
            CASE(1, "s_nop 0\n")
            CASE(2, "s_nop 1\n")
            CASE(3, "s_nop 2\n")
            CASE(4, "s_nop 3\n")
            CASE(5, "s_nop 4\n")
            CASE(6, "s_lshr_b64 s[86:87], s[84:85], 16\n")
            CASE(7, "s_lshr_b64 s[86:87], s[84:85], 16\n s_nop 0\n")
            CASE(8, "v_mov_b32 v59, v58\n")
            CASE(9, "v_mov_b32 v59, v58\n v_mov_b32 v57, v58\n")
        default: a = t = 0;
        }
        if ((it & 63) == (int)lane) {
            res_a = a;
            res_v = t;
        }
        if ((it & 63) == 63) {
            out_a[it - 63 + lane] = res_a;
            out_v[it - 63 + lane] = res_v;
        }
    }
}
int main()
{
    unsigned hg[64], *g, *oa, *ov;
    for (int i = 0; i < 64; ++i) hg[i] = i < 40 ? 60000u - 1500u * i : 0u;  // decreasing, zero pad behind lane 39
    hipMalloc(&g, 256);
    hipMalloc(&oa, N * 4);
    hipMalloc(&ov, N * 4);
    hipMemcpy(g, hg, 256, hipMemcpyHostToDevice);
    for (int nops = 0; nops < 10; ++nops) {
        hipMemset(oa, 0xff, N * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, oa, ov, nops);
        if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 1; }
        static unsigned ha[N], hv[N];
        hipMemcpy(ha, oa, N * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hv, ov, N * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int it = 0; it < N; ++it) {
            const unsigned s = (it * 37u) & 0xFFFFu;
            unsigned e = 0;
            while (!(s >= hg[e])) ++e;
            if (ha[it] != e || hv[it] != hg[e]) {
                if (bad < 0) printf("  it %d s %u: got lane %u value %u, expected lane %u value %u\n", it, s, ha[it], hv[it], e, hg[e]);
                ++bad;
            }
        }
        printf("variant %d: %d of %d wrong\n", nops, bad, N);
    }
    return 0;
}
