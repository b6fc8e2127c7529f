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
