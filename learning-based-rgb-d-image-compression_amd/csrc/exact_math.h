// Device restatements of the CPU math the reference's float path goes through (torch 2.10 CPU, AVX-512 build).
// They reproduce those kernels' results bit for bit: every operation below is an IEEE fp32 operation in the order the CPU
// code performs it, with fused multiply-adds exactly where that code has them.
#pragma once
#include <hip/hip_runtime.h>

// exp(x) as Sleef's expf_u10 computes it (the vector exp behind at::vec::Vectorized<float>::exp(), which torch's sigmoid kernel
// calls; Sleef is a third-party dependency of torch, algorithm restated from its published description: Cody-Waite reduction by
// ln 2 in two parts, degree-5 polynomial evaluated with fused multiply-adds, scaling by 2^q in two steps).
__device__ __forceinline__ float rgbd_expf_u10(float d)
{
    const float qf = rintf(__fmul_rn(d, 1.442695040888963407359924681001892137426645954152985934135449406931f));
    const int q = (int)qf;
    float s = __fmaf_rn(qf, -0.693145751953125f, d);
    s = __fmaf_rn(qf, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __fmaf_rn(u, s, 0.00139304355252534151077271f);
    u = __fmaf_rn(u, s, 0.00833336077630519866943359f);
    u = __fmaf_rn(u, s, 0.0416664853692054748535156f);
    u = __fmaf_rn(u, s, 0.166666671633720397949219f);
    u = __fmaf_rn(u, s, 0.5f);
    u = __fadd_rn(1.0f, __fmaf_rn(__fmul_rn(s, s), u, s));
    const int q1 = q >> 1, q2 = q - q1;
    u = __fmul_rn(__fmul_rn(u, __int_as_float((q1 + 127) << 23)), __int_as_float((q2 + 127) << 23));
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = __int_as_float(0x7f800000);
    return u;
}

// torch.sigmoid on the CPU's vector path: 1 / (1 + exp(0 - x)), IEEE division
__device__ __forceinline__ float rgbd_sigmoid_ref(float x)
{
    return __fdiv_rn(1.0f, __fadd_rn(1.0f, rgbd_expf_u10(__fsub_rn(0.0f, x))));
}

// expf as the C library computes it (glibc's e_expf.c = the published ARM optimized-routines algorithm: x * 32 / ln 2 = k + r,
// exp(x) = 2^(k/32) * p(r) in double, one rounding to float; table = bits(2^(i/32)) - (i << 47)).  It is what torch.sigmoid's
// SCALAR path calls (std::exp); oracle/cpu_arith.c: orc_expf_libm is the same code and is checked against libm.
static __device__ const unsigned long long rgbd_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
__device__ __forceinline__ float rgbd_expf_libm(float x)
{
    if (x != x) return x;
    if (x > 88.72283f) return __int_as_float(0x7f800000);
    if (x < -103.97208f) return 0.0f;
    const double z0 = __dmul_rn(0x1.71547652b82fep+0 * 32, (double)x);
    double kd = __dadd_rn(z0, 0x1.8p+52);
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd = __dsub_rn(kd, 0x1.8p+52);
    const double r = __dsub_rn(z0, kd);
    const double sc = __longlong_as_double((long long)(rgbd_exp2f_tab[ki & 31] + (ki << 47)));
    const double z = __dadd_rn(__dmul_rn(0x1.c6af84b912394p-5 / 32 / 32 / 32, r), 0x1.ebfce50fac4f3p-3 / 32 / 32);
    const double r2 = __dmul_rn(r, r);
    double y = __dadd_rn(__dmul_rn(0x1.62e42ff0c52d6p-1 / 32, r), 1.0);
    y = __dadd_rn(__dmul_rn(z, r2), y);
    return __double2float_rn(__dmul_rn(y, sc));
}

// Is element i of a contiguous tensor of n elements handled by the SCALAR tail of an ATen vectorised elementwise loop run by
// `threads` CPU threads?  (TensorIterator::for_each: serial below 32768 elements; else at::parallel_for: min(threads,
// ceil(n / 32768)) tasks of ceil(n / tasks) elements; inside a range cpu/Loops.h vectorized_loop takes 32 elements per step
// and the last len % 32 go through the scalar op.)  oracle/cpu_arith.c: orc_aten_scalar_tail.
__device__ __forceinline__ bool rgbd_aten_scalar_tail(long i, long n, int threads)
{
    long tasks = 1;
    if (n >= 32768 && threads > 1) {
        tasks = (n + 32767) / 32768;
        if (tasks > threads) tasks = threads;
    }
    const long chunk = (n + tasks - 1) / tasks, c0 = (i / chunk) * chunk;
    const long len = (n - c0 < chunk) ? n - c0 : chunk;
    return (i - c0) >= len - (len % 32);
}

// torch.sigmoid's scalar path: 1 / (1 + std::exp(-x))
__device__ __forceinline__ float rgbd_sigmoid_ref_scalar(float x)
{
    return __fdiv_rn(1.0f, __fadd_rn(1.0f, rgbd_expf_libm(-x)));
}
