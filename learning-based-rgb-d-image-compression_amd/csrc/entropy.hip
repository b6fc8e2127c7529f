// Entropy stage of the ELIC_united path on gfx950: checkerboard quantise/index/scatter (utils/ckbd.py:83-125,
// entropy_models.py:118-146,561-568), the factorised-prior z path (entropy_models.py:195-266,431-446) and the
// 64-bit-state rANS coder (cpp_exts/rans/rans_interface.cpp:99-351 + third_party/ryg_rans/rans64.h:59-142).
//
// Symbols, indexes and bitstreams stay in HBM: the reference's 44 device<->host round trips per image disappear.
// The coder is integer work and bit-exact against the oracle by construction; streams are little-endian u32 words.
// One wavefront serves one stream: all 64 lanes expand symbols to (start, freq) pairs / fetch table metadata for a
// chunk in parallel, then lane 0 walks the serial state recurrence out of LDS.
#include <stdlib.h>

#include <mutex>

#include "common.h"

#define PROB_BITS 16
#define ESC_BITS 4
#define ESC_MAX 15u
#define RANS_LOW (1ull << 31)


// (ckbd_col / scale_to_index: common.h -- shared with the C ABI's stand-alone checkerboard operators, coder_abi.hip)
__device__ __forceinline__ int64_t sym_pos(const PartGeom& g, const int64_t* stream_base, int64_t part_off, int b,
                                           int c, int row, int k)
{
    const int64_t w2 = g.w / 2;
    if (g.per_image) return stream_base[b] + part_off + ((int64_t)c * g.h + row) * w2 + k;
    return stream_base[0] + part_off * g.B + (((int64_t)b * g.C + c) * g.h + row) * w2 + k;
}

// mode 0: encode (quantise + index + scatter yhat), 1: index only, 2: decode (yhat = sym + mean)
template <int MODE>
__global__ void ckbd_part_kernel(const float* __restrict__ y, int ycs, const float* __restrict__ params, int pcs,
                                 float* __restrict__ yhat, int yhcs, const float* __restrict__ table, PartGeom g,
                                 int32_t* __restrict__ sym, int32_t* __restrict__ idx,
                                 const int64_t* __restrict__ stream_base, int64_t part_off,
                                 float* __restrict__ dbg_x = nullptr, float* __restrict__ dbg_s = nullptr)
{
    __shared__ float tbl[64];
    if (threadIdx.x < 64) tbl[threadIdx.x] = table[threadIdx.x];
    __syncthreads();
    const int w2 = g.w / 2;
    const size_t total = (size_t)g.B * g.h * w2 * g.C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % g.C);
        size_t t = i / g.C;
        const int k = (int)(t % w2);
        t /= w2;
        const int row = (int)(t % g.h);
        const int b = (int)(t / g.h);
        const int col = ckbd_col(row, k, g.anchor);
        const size_t pix = ((size_t)b * g.h + row) * g.w + col;
        const int pc = rgbd_cperm(c, g.perm);  // where channel c sits in the tensors (g.C is a multiple of 16 when perm)
        const float scale = params[pix * pcs + pc];
        const float mean = params[pix * pcs + g.C + pc];
        const int64_t pos = sym_pos(g, stream_base, part_off, b, c, row, k);
        if (MODE == 0) {
            const float xm = y[pix * ycs + pc] - mean;
            const float r = rintf(xm);  // round half to even, like torch.round
            const int s = (int)r;
            if (dbg_x) {  // parity bookkeeping (rgbd_elic_set_debug_floats): the rounded value and the indexed scale
                dbg_x[pos] = xm;
                dbg_s[pos] = scale;
            }
            sym[pos] = s;
            idx[pos] = scale_to_index(tbl, scale);
            yhat[pix * yhcs + pc] = (float)s + mean;
        } else if (MODE == 1) {
            idx[pos] = scale_to_index(tbl, scale);
        } else {
            yhat[pix * yhcs + pc] = (float)sym[pos] + mean;
        }
        if (MODE != 1 && g.anchor) {
            // the anchor pass defines the whole slice: off-parity positions start at zero (ckbd.py:66-72)
            const size_t opix = ((size_t)b * g.h + row) * g.w + (col ^ 1);
            yhat[opix * yhcs + pc] = 0.f;
        }
    }
}

static inline unsigned part_grid(const PartGeom& g)
{
    size_t work = (size_t)g.B * g.h * (g.w / 2) * g.C;
    size_t n = (work + 255) / 256;
    return (unsigned)(n < 1 ? 1 : (n > 2048 ? 2048 : n));
}

int launch_ckbd_encode_part(const float* y, int ycs, const float* params, int pcs, float* yhat, int yhcs,
                            const float* table, PartGeom g, int32_t* sym, int32_t* idx, const int64_t* stream_base,
                            int64_t part_off, hipStream_t s, float* dbg_x, float* dbg_s)
{
    if (g.w % 2) return RGBD_EINVAL;
    hipLaunchKernelGGL(ckbd_part_kernel<0>, dim3(part_grid(g)), dim3(256), 0, s, y, ycs, params, pcs, yhat, yhcs, table,
                       g, sym, idx, stream_base, part_off, dbg_x, dbg_s);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

int launch_ckbd_index_part(const float* params, int pcs, const float* table, PartGeom g, int32_t* idx,
                           const int64_t* stream_base, int64_t part_off, hipStream_t s)
{
    if (g.w % 2) return RGBD_EINVAL;
    hipLaunchKernelGGL(ckbd_part_kernel<1>, dim3(part_grid(g)), dim3(256), 0, s, (const float*)nullptr, 0, params, pcs,
                       (float*)nullptr, 0, table, g, (int32_t*)nullptr, idx, stream_base, part_off, (float*)nullptr, (float*)nullptr);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

int launch_ckbd_decode_part(const float* params, int pcs, float* yhat, int yhcs, PartGeom g, const int32_t* sym,
                            const int64_t* stream_base, int64_t part_off, hipStream_t s)
{
    if (g.w % 2) return RGBD_EINVAL;
    hipLaunchKernelGGL(ckbd_part_kernel<2>, dim3(part_grid(g)), dim3(256), 0, s, (const float*)nullptr, 0, params, pcs,
                       yhat, yhcs, params /*unused table*/, g, const_cast<int32_t*>(sym), (int32_t*)nullptr, stream_base,
                       part_off, (float*)nullptr, (float*)nullptr);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// ---------------------------------------------------------------------------------------------
// eval-mode forward() (models/elic_united.py:94-115, 234-263): same quantisation as the encoder, plus the Gaussian
// likelihood of the quantised value (entropy_models.py:534-558) instead of symbols
__global__ void ckbd_estimate_kernel(const float* __restrict__ y, int ycs, const float* __restrict__ params, int pcs,
                                     float* __restrict__ yhat, int yhcs, float* __restrict__ lik, int lcs, PartGeom g)
{
    const int w2 = g.w / 2;
    const size_t total = (size_t)g.B * g.h * w2 * g.C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % g.C);
        size_t t = i / g.C;
        const int k = (int)(t % w2);
        t /= w2;
        const int row = (int)(t % g.h);
        const int b = (int)(t / g.h);
        const int col = ckbd_col(row, k, g.anchor);
        const size_t pix = ((size_t)b * g.h + row) * g.w + col;
        const int pc = rgbd_cperm(c, g.perm);
        const float scale = fmaxf(params[pix * pcs + pc], 0.11f);  // LowerBound(0.11)
        const float mean = params[pix * pcs + g.C + pc];
        const float out = __fadd_rn(rintf(y[pix * ycs + pc] - mean), mean);  // quantize(..., "dequantize", means)
        yhat[pix * yhcs + pc] = out;
        const float v = fabsf(__fsub_rn(out, mean));
        const float cst = -0.70710678118654752440f;  // -(2 ** -0.5)
        const float upper = 0.5f * erfcf(cst * ((0.5f - v) / scale));
        const float lower = 0.5f * erfcf(cst * ((-0.5f - v) / scale));
        lik[pix * lcs + pc] = fmaxf(upper - lower, 1e-9f);
        if (g.anchor) {
            const size_t opix = ((size_t)b * g.h + row) * g.w + (col ^ 1);
            yhat[opix * yhcs + pc] = 0.f;
        }
    }
}

int launch_ckbd_estimate_part(const float* y, int ycs, const float* params, int pcs, float* yhat, int yhcs, float* lik,
                              int lcs, PartGeom g, hipStream_t s)
{
    if (g.w % 2) return RGBD_EINVAL;
    hipLaunchKernelGGL(ckbd_estimate_kernel, dim3(part_grid(g)), dim3(256), 0, s, y, ycs, params, pcs, yhat, yhcs, lik, lcs,
                       g);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// Factorised prior in eval mode (entropy_models.py:369-428): z_hat = round(z - median) + median and its likelihood
// |sigmoid(s*u) - sigmoid(s*l)| from the per-channel cumulative (a 1-3-3-3-3-1 network).  `prm` holds, per channel,
// softplus(matrix_i) / bias_i / tanh(factor_i) flattened in layer order: 3+3+3, (9+3+3)x3, 3+1  = 58 floats.
__device__ __forceinline__ float eb_logits(const float* __restrict__ p, float x)
{
    float h[3], t[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float l = p[j] * x + p[3 + j];
        h[j] = l + p[6 + j] * tanhf(l);
    }
    p += 9;
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float l = p[j * 3] * h[0];
            l += p[j * 3 + 1] * h[1];
            l += p[j * 3 + 2] * h[2];
            l += p[9 + j];
            t[j] = l + p[12 + j] * tanhf(l);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) h[j] = t[j];
        p += 15;
    }
    float l = p[0] * h[0];
    l += p[1] * h[1];
    l += p[2] * h[2];
    return l + p[3];
}

__global__ void eb_forward_kernel(const float* __restrict__ z, int zcs, int B, int h, int w, int C,
                                  const float* __restrict__ med, const float* __restrict__ prm, float* __restrict__ zhat,
                                  float* __restrict__ lik, int perm)
{
    const size_t total = (size_t)B * h * w * zcs;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pc = (int)(i % zcs);           // position in the tensor
        const int c = rgbd_cperm(pc, perm);      // the channel stored there (the permutation is its own inverse)
        const size_t pix = i / zcs;
        if (c >= C) {
            zhat[pix * zcs + pc] = 0.f;
            lik[pix * zcs + pc] = 0.f;
            continue;
        }
        const float out = __fadd_rn(rintf(z[pix * zcs + pc] - med[c]), med[c]);
        zhat[pix * zcs + pc] = out;
        const float lo = eb_logits(prm + (size_t)c * 58, out - 0.5f);
        const float up = eb_logits(prm + (size_t)c * 58, out + 0.5f);
        const float sm = lo + up;
        const float sg = sm > 0.f ? -1.f : (sm < 0.f ? 1.f : 0.f);
        const float a = 1.0f / (1.0f + expf(-(sg * up))), b = 1.0f / (1.0f + expf(-(sg * lo)));
        lik[pix * zcs + pc] = fmaxf(fabsf(a - b), 1e-9f);
    }
}

int launch_eb_forward(const float* z, int zcs, int B, int h, int w, int C, const float* med, const float* prm, float* zhat,
                      float* lik, hipStream_t s, int perm)
{
    const size_t work = (size_t)B * h * w * zcs;
    hipLaunchKernelGGL(eb_forward_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, z, zcs, B, h, w, C, med, prm,
                       zhat, lik, perm);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// ---------------------------------------------------------------------------------------------
// z path: sym = round(z - median_c), index = c, per-image streams in (c, row, col) order
__global__ void z_quant_kernel(const float* __restrict__ z, int zcs, int B, int h, int w, int C,
                               const float* __restrict__ med, int32_t* __restrict__ sym, int32_t* __restrict__ idx, int perm)
{
    const size_t total = (size_t)B * h * w * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t pix = i / C;
        const size_t hw = pix % ((size_t)h * w);
        const size_t b = pix / ((size_t)h * w);
        const size_t pos = (b * C + c) * (size_t)h * w + hw;
        sym[pos] = (int)rintf(z[pix * zcs + rgbd_cperm(c, perm)] - med[c]);
        idx[pos] = c;
    }
}

int launch_z_quant(const float* z, int zcs, int B, int h, int w, int C, const float* medians, int32_t* sym, int32_t* idx,
                   hipStream_t s, int perm)
{
    const size_t work = (size_t)B * h * w * C;
    hipLaunchKernelGGL(z_quant_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, z, zcs, B, h, w, C, medians,
                       sym, idx, perm);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

__global__ void z_dequant_kernel(const int32_t* __restrict__ sym, int B, int h, int w, int C,
                                 const float* __restrict__ med, float* __restrict__ zhat, int zcs, int perm)
{
    const size_t total = (size_t)B * h * w * zcs;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pc = (int)(i % zcs);
        const int c = rgbd_cperm(pc, perm);  // the channel stored at position pc
        const size_t pix = i / zcs;
        const size_t hw = pix % ((size_t)h * w);
        const size_t b = pix / ((size_t)h * w);
        float v = 0.f;
        if (c < C) v = (float)sym[(b * C + c) * (size_t)h * w + hw] + med[c];
        zhat[pix * zcs + pc] = v;
    }
}

int launch_z_dequant(const int32_t* sym, int B, int h, int w, int C, const float* medians, float* zhat, int zcs,
                     hipStream_t s, int perm)
{
    const size_t work = (size_t)B * h * w * zcs;
    hipLaunchKernelGGL(z_dequant_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, sym, B, h, w, C, medians,
                       zhat, zcs, perm);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// ---------------------------------------------------------------------------------------------
// rANS coder kernels.  The state recurrence of one stream is strictly serial, and one wavefront issues at most one
// instruction every ~4 clocks, so the design goal is the fewest instructions and the fewest LDS/memory round trips per
// symbol on the critical path:
//   * the recurrence runs in SGPRs (every lane of wave 0 executes the same uniform code; branches are scalar);
//   * everything that does not depend on the state is produced 64 symbols at a time by the 64 lanes in parallel
//     (coalesced loads, table gathers, fp64 reciprocals) and handed to the scalar chain with v_readlane;
//   * results go back with v_writelane and leave as one coalesced 256-byte store per 64 items;
//   * the decoder's only dependent memory access per symbol is one 8-byte LDS read of the bucket table.
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t rfl64(uint64_t v)
{
    return (uint64_t)rfl((uint32_t)v) | ((uint64_t)rfl((uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ uint32_t rdl(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
// lane `sel` of the result takes the (uniform) value, the others keep theirs: v_cmp + v_cndmask
__device__ __forceinline__ uint32_t wrl(uint32_t val, int sel, uint32_t old)
{
    return ((int)(threadIdx.x & 63) == sel) ? val : old;
}

// ---------------------------------------------------------------------------------------------
// Encoder: one wavefront per stream, symbols consumed from the back (rans_interface.cpp:167-185 pops from the back).
struct EncBatch {
    uint32_t mlo, mhi;  // reciprocal multiplier of the symbol's frequency (DevTables::enc)
    uint32_t w2;        // bias | shift << 17 | escape << 31
    uint32_t freq;
    uint32_t raw;       // escape payload
};

// The per-symbol operands are produced 64 at a time by the lanes, two batches ahead of the serial loop and in two steps,
// so that neither of the two dependent global loads (symbol/index, then the symbol's table entry) is ever waited for:
//   enc_load   (batch b-2): symbol and table index from HBM
//   enc_finish (batch b-1): row geometry from LDS, escape split, gather of the reciprocal entry
struct EncRaw {
    int ti, sv;
    bool valid;
};

__device__ __forceinline__ EncRaw enc_load(const int32_t* __restrict__ sym, const int32_t* __restrict__ idx, int64_t pos,
                                           bool valid)
{
    EncRaw r;
    r.valid = valid;
    r.ti = valid ? idx[pos] : 0;
    r.sv = valid ? sym[pos] : 0;
    return r;
}

__device__ __forceinline__ EncBatch enc_finish(const DevTables& t, const int3* __restrict__ rowmeta, const EncRaw& r)
{
    EncBatch b;
    b.mlo = b.mhi = ~0u;
    b.w2 = 65535u;
    b.freq = 1u;
    b.raw = 0;
    if (r.valid) {
        const int3 rm = rowmeta[r.ti];  // {row_off, cdf_length, offset}
        const int top = rm.y - 2;
        int v = r.sv - rm.z;
        uint32_t raw = 0;
        if (v < 0) {
            raw = (uint32_t)(-2 * v - 1);
            v = top;
        } else if (v >= top) {
            raw = (uint32_t)(2 * (v - top));
            v = top;
        }
        const uint4 e = reinterpret_cast<const uint4*>(t.enc)[rm.x + v];
        b.mlo = e.x;
        b.mhi = e.y;
        b.w2 = e.z | ((v == top) ? 0x80000000u : 0u);
        b.freq = e.w;
        b.raw = raw;
    }
    return b;
}

__global__ __launch_bounds__(64) void rans_encode_kernel(const int32_t* __restrict__ sym, const int32_t* __restrict__ idx,
                                                         const int64_t* __restrict__ sym_base,
                                                         const int64_t* __restrict__ counts, int split, DevTables t0,
                                                         DevTables t1, uint32_t* __restrict__ out, int64_t cap_words,
                                                         int64_t* __restrict__ out_words, int* __restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char esm[];
    int3* rowmeta = reinterpret_cast<int3*>(esm);
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    const DevTables& t = s < split ? t0 : t1;
    const int64_t n = counts[s];
    const int64_t base = sym_base[s];
    uint32_t* o = out + (size_t)s * cap_words;  // cap_words is a multiple of 64 (and < 2^31: checked by the launcher)
    uint64_t x = RANS_LOW;
    int w = (int)cap_words;  // next free slot is w-1; slot k lives in lane (k & 63) of `ov` until its 64-block is full
    uint32_t ov = 0;
    int bad = 0;

    auto emit = [&]() {  // escape path only; the symbol loop below has its own copy in ISA
        --w;
        ov = wrl((uint32_t)x, w & 63, ov);
        x >>= 32;
        if ((w & 63) == 0) o[w + lane] = ov;  // block [w, w+64) complete: one coalesced 256-byte store
    };

    for (int i = lane; i < t.nrows; i += 64) rowmeta[i] = make_int3(t.row_off[i], t.sizes[i], t.offsets[i]);
    __syncthreads();
    const int64_t nb = (n + 63) >> 6;
    EncBatch cur = enc_finish(t, rowmeta, enc_load(sym, idx, base + (nb - 1) * 64 + lane, nb > 0 && (nb - 1) * 64 + lane < n));
    EncRaw raw1 = enc_load(sym, idx, base + (nb - 2) * 64 + lane, nb > 1);
    for (int64_t b = nb - 1; b >= 0; --b) {
        const EncBatch nxt = enc_finish(t, rowmeta, raw1);                   // batch b-1: its symbols arrived last round
        raw1 = enc_load(sym, idx, base + (b - 2) * 64 + lane, b > 1);         // batch b-2: in flight during this round
        const int cnt = (int)((n - b * 64) < 64 ? (n - b * 64) : 64);
        if (w < 64 + 10 * 64) {  // worst case for one batch: 64 * (1 + 1 + 8) items
            bad = 1;
            break;
        }
        int j = cnt - 1;
        while (j >= 0) {
            // The symbol loop in ISA; the state lives in a VGPR pair (all lanes hold the same value) because the exact
            // division is four v_mad_u64_u32:  q = mulhi64(x, m) >> shift,  x = x + bias + q * (65536 - freq)
            // (rans64.h:77-94 with the reciprocal of rans64.h:167-278).  It leaves the loop for the two rare events:
            //   code 1: symbol j carries an escape payload (coded below, then the loop resumes at j)
            //   code 2: a 64-word output block is complete and wants storing
            // v[60:61] = x, v[62:67] scratch (v65 stays 0).  VALU->VALU reads of vcc need 2 wait states on gfx950.
            uint32_t code, w2, fr, ml, mh, tt, t2, t3, t4, m0s;
            j = (int)rfl((uint32_t)j);
            w = (int)rfl((uint32_t)w);
            asm volatile(
                "v_mov_b64 v[60:61], %[x]\n"
                "v_mov_b32 v65, 0\n"
                "s_mov_b32 %[m0s], m0\n"                  // m0 is reserved: borrowed for the lane index, restored below
                "s_mov_b32 m0, %[j]\n"
                "1:\n"
                "v_readlane_b32 %[w2], %[vw2], m0\n"
                "v_readlane_b32 %[fr], %[vfr], m0\n"
                "v_readlane_b32 %[ml], %[vml], m0\n"
                "v_readlane_b32 %[mh], %[vmh], m0\n"
                "s_cmp_lt_i32 %[w2], 0\n"
                "s_cbranch_scc1 7f\n"
                "10:\n"
                "v_lshrrev_b64 v[62:63], 47, v[60:61]\n"   // x >= freq << 47 ?  (rans64.h:82-83)
                "v_cmp_le_u32 vcc, %[fr], v62\n"
                "s_cbranch_vccz 2f\n"
                "s_sub_u32 %[w], %[w], 1\n"                // emit the low word into slot w-1
                "s_and_b32 %[tt], %[w], 63\n"
                "v_cmp_eq_u32 vcc, %[tt], %[lane]\n"
                "s_nop 1\n"
                "v_cndmask_b32 %[ov], %[ov], v60, vcc\n"
                "v_mov_b32 v60, v61\n"
                "v_mov_b32 v61, 0\n"
                "s_cmp_eq_u32 %[tt], 0\n"
                "s_cbranch_scc1 8f\n"
                "2:\n"
                "v_mad_u64_u32 v[62:63], vcc, v60, %[ml], 0\n"           // xl*ml
                "v_lshrrev_b64 v[62:63], 32, v[62:63]\n"
                "v_mad_u64_u32 v[62:63], vcc, v61, %[ml], v[62:63]\n"    // xh*ml + hi
                "v_mov_b32 v64, v62\n"
                "v_mad_u64_u32 v[66:67], vcc, v60, %[mh], v[64:65]\n"    // xl*mh + lo
                "v_mov_b32 v64, v63\n"
                "v_mad_u64_u32 v[62:63], vcc, v61, %[mh], v[64:65]\n"    // xh*mh + hi
                "v_mad_u64_u32 v[62:63], vcc, v67, 1, v[62:63]\n"        // + hi          = mulhi64(x, m)
                "s_bfe_u32 %[tt], %[w2], 0x50011\n"                       // shift = (w2 >> 17) & 31
                "v_lshrrev_b64 v[62:63], %[tt], v[62:63]\n"              // q
                "s_and_b32 %[w2], %[w2], 0x1ffff\n"                       // bias
                "s_sub_u32 %[fr], 0x10000, %[fr]\n"                       // 65536 - freq
                "v_mad_u64_u32 v[60:61], vcc, %[w2], 1, v[60:61]\n"      // x += bias
                "v_mad_u64_u32 v[60:61], vcc, v62, %[fr], v[60:61]\n"    // x += q.lo * (65536 - freq)
                "v_mad_u32_u24 v61, v63, %[fr], v61\n"                    // x.hi += q.hi * (65536 - freq)   (q < 2^47)
                "s_sub_u32 m0, m0, 1\n"
                "s_cmp_ge_i32 m0, 0\n"
                "s_cbranch_scc1 1b\n"
                "s_mov_b32 %[code], 0\n"
                "s_branch 9f\n"
                // escape payload (reverse of rans_interface.cpp:147-162): the value's nibbles, then their count, pushed with
                // 4-bit steps; afterwards the escape slot itself is coded like any symbol (label 10).  At most two words are
                // emitted here and one more by the slot itself, so with four free slots in the current output block no store
                // becomes due before the symbol is finished (a store exit re-enters at the symbol's start).
                "7:\n"
                "s_and_b32 %[tt], %[w], 63\n"
                "s_cmp_lt_u32 %[tt], 4\n"
                "s_cbranch_scc1 70f\n"
                "v_readlane_b32 %[t4], %[vraw], m0\n"
                "s_flbit_i32_b32 %[t2], %[t4]\n"
                "s_sub_u32 %[t2], 35, %[t2]\n"
                "s_lshr_b32 %[t2], %[t2], 2\n"             // nibbles = ceil(bits / 4)
                "s_cmp_eq_u32 %[t4], 0\n"
                "s_cselect_b32 %[t2], 0, %[t2]\n"
                "s_mov_b32 %[t3], %[t2]\n"
                "71:\n"
                "s_cmp_eq_u32 %[t3], 0\n"
                "s_cbranch_scc1 72f\n"
                "s_sub_u32 %[t3], %[t3], 1\n"
                "s_lshl_b32 %[tt], %[t3], 2\n"
                "s_lshr_b32 %[code], %[t4], %[tt]\n"
                "s_and_b32 %[code], %[code], 15\n"
                "v_lshrrev_b64 v[62:63], 59, v[60:61]\n"   // x >= 2^59: emit first (rans_interface.cpp:67-68)
                "v_cmp_ne_u32 vcc, 0, v62\n"
                "s_cbranch_vccz 73f\n"
                "s_sub_u32 %[w], %[w], 1\n"
                "s_and_b32 %[tt], %[w], 63\n"
                "v_cmp_eq_u32 vcc, %[tt], %[lane]\n"
                "s_nop 1\n"
                "v_cndmask_b32 %[ov], %[ov], v60, vcc\n"
                "v_mov_b32 v60, v61\n"
                "v_mov_b32 v61, 0\n"
                "73:\n"
                "v_lshlrev_b64 v[60:61], 4, v[60:61]\n"
                "v_or_b32 v60, %[code], v60\n"
                "s_branch 71b\n"
                "72:\n"
                "v_lshrrev_b64 v[62:63], 59, v[60:61]\n"   // x >= 2^59: emit first (rans_interface.cpp:67-68)
                "v_cmp_ne_u32 vcc, 0, v62\n"
                "s_cbranch_vccz 73f\n"
                "s_sub_u32 %[w], %[w], 1\n"
                "s_and_b32 %[tt], %[w], 63\n"
                "v_cmp_eq_u32 vcc, %[tt], %[lane]\n"
                "s_nop 1\n"
                "v_cndmask_b32 %[ov], %[ov], v60, vcc\n"
                "v_mov_b32 v60, v61\n"
                "v_mov_b32 v61, 0\n"
                "73:\n"
                "v_lshlrev_b64 v[60:61], 4, v[60:61]\n"
                "v_or_b32 v60, %[t2], v60\n"
                "s_branch 10b\n"
                "70:\n"
                "s_mov_b32 %[code], 1\n"
                "s_branch 9f\n"
                "8:\n"
                "s_mov_b32 %[code], 2\n"
                "9:\n"
                "s_mov_b32 %[j], m0\n"
                "s_mov_b32 m0, %[m0s]\n"
                "v_mov_b64 %[x], v[60:61]\n"
                : [x] "+v"(x), [j] "+s"(j), [w] "+s"(w), [ov] "+v"(ov), [code] "=&s"(code), [w2] "=&s"(w2), [fr] "=&s"(fr),
                  [ml] "=&s"(ml), [mh] "=&s"(mh), [tt] "=&s"(tt), [t2] "=&s"(t2), [t3] "=&s"(t3), [t4] "=&s"(t4),
                  [m0s] "=&s"(m0s)
                : [vw2] "v"(cur.w2), [vfr] "v"(cur.freq), [vml] "v"(cur.mlo), [vmh] "v"(cur.mhi), [lane] "v"(lane),
                  [vraw] "v"(cur.raw)
                : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "vcc", "scc", "memory");
            x = rfl64(x);
            code = rfl(code);
            j = (int)rfl((uint32_t)j);
            w = (int)rfl((uint32_t)w);
            if (code == 0) break;
            if (code == 2) {  // block [w, w+64) complete: one coalesced 256-byte store
                o[w + lane] = ov;
                continue;
            }
            // escape payload first (reverse of rans_interface.cpp:147-162), then the loop codes the escape slot itself
            const uint32_t raw = rdl(cur.raw, j);
            int nn = 0;
            while (nn < 8 && (raw >> (nn * ESC_BITS)) != 0) ++nn;
            for (int k = nn - 1; k >= -1; --k) {
                const uint32_t val = k >= 0 ? ((raw >> (k * ESC_BITS)) & ESC_MAX) : (uint32_t)nn;
                if (x >= (1ull << 59)) emit();  // ((RANS_LOW >> 16) << 32) << 12, rans_interface.cpp:67-68
                x = (x << ESC_BITS) | val;
            }
            cur.w2 = wrl(rdl(cur.w2, j) & 0x7FFFFFFFu, j, cur.w2);  // payload done: lane j is an ordinary symbol now
        }
        cur = nxt;
    }
    if (bad) {
        if (lane == 0) {
            *err = 1;
            out_words[s] = 0;
        }
        return;
    }
    // rans64.h:96-103: the two state words go in front
    --w;
    ov = wrl((uint32_t)(x >> 32), w & 63, ov);
    if ((w & 63) == 0) o[w + lane] = ov;
    --w;
    ov = wrl((uint32_t)x, w & 63, ov);
    if ((w & 63) == 0) o[w + lane] = ov;
    else if (lane >= (w & 63)) o[(w & ~63) + lane] = ov;  // partial leading block
    if (lane == 0) out_words[s] = cap_words - w;
}

// ---------------------------------------------------------------------------------------------
// Encoder, second / third generation: the same stream bit for bit, with a branch-free serial loop.
//
// Measured on gfx950 (tools/ubench/oplat.hip): one dependent ALU instruction of a single wave costs ~3.5 ns, an
// independent one ~2.3 ns of issue time, but every branch costs 8-17 ns (s_cbranch_vccz after a v_cmp: 17 ns) -- the
// first-generation loop above spends a third of its 137 ns per symbol in its four branches.  Here
//   * escapes are expanded into ITEMS beforehand, in parallel by the 64 lanes: an item is either a table symbol or a raw
//     4-bit nibble (rans_interface.cpp:60-78,147-162), and both are the same arithmetic
//         renormalise when (x >> 47) >= thr (items hold thr << 15, compared with x's high word);
//         x = x + bias + (mulhi64(x, m) >> shift) * mult
//     (symbol: thr = freq, mult = 65536 - freq, m the exact reciprocal of rans64.h:167-278; nibble v: thr = 4096 (x >= 2^59),
//      m = 2^64 - 1 so q = x - 1, mult = 15, bias = v + 15, i.e. x = (x << 4) | v), so the serial loop has no escape path;
//   * the renormalisation is branch-free: the candidate word is ALWAYS written to an LDS staging slot and the slot
//     pointer moves only when the word was due (v_cndmask), the state halves are selected with v_cndmask;
//   * items are read from LDS one item ahead (two register sets, loop unrolled four times), nothing is fetched with
//     v_readlane, and the loop runs with a single active lane (all of it is uniform work).
// Per batch of 64 symbols the staged words are copied out with one coalesced store.
#define ENC_ITEM_CAP 704  // 64 symbols x (1 slot + 1 count + 8 payload nibbles) + padding
#define ENC_ITEM(S0, S1, S2, S3, S4, S5)                                                            \
    "ds_write_b32 %[wa], v60\n"                      /* candidate word (kept only if due) */        \
    "v_cmp_ge_u32 vcc, v61, " S5 "\n"                /* x.hi >= thr << 15, i.e. (x >> 47) >= thr: renormalise */ \
    "v_cndmask_b32 v60, v60, v61, vcc\n"                                                            \
    "v_cndmask_b32 v61, v61, v84, vcc\n"             /* v84 = 0 */                                   \
    "v_cndmask_b32 v63, 0, v85, vcc\n"               /* v85 = 4 */                                   \
    "v_sub_u32 %[wa], %[wa], v63\n"                                                                 \
    "v_mul_hi_u32 v86, v60, " S0 "\n"                             /* A.hi = hi32(xl * ml); v87 = 0 */ \
    "v_mad_u64_u32 v[82:83], s[94:95], " S2 ", 1, v[60:61]\n"     /* xb = x + bias */                \
    "v_mad_u64_u32 v[64:65], s[94:95], v61, " S0 ", v[86:87]\n"   /* B = xh * ml + A.hi */           \
    "v_mov_b32 v88, v64\n"                                        /* {B.lo, 0}: v89 = 0 */           \
    "v_mov_b32 v90, v65\n"                                        /* {B.hi, 0}: v91 = 0 */           \
    "v_mad_u64_u32 v[66:67], s[94:95], v60, " S1 ", v[88:89]\n"   /* C = xl * mh + B.lo */           \
    "v_mad_u64_u32 v[68:69], s[94:95], v61, " S1 ", v[90:91]\n"   /* D = xh * mh + B.hi */           \
    "v_mad_u64_u32 v[68:69], s[94:95], v67, 1, v[68:69]\n"        /* + C.hi = mulhi64(x, m) */       \
    "v_lshrrev_b64 v[68:69], " S3 ", v[68:69]\n"                  /* q */                            \
    "v_mul_lo_u32 v64, v69, " S4 "\n"                             /* q.hi * mult (nibble items: q ~ x) */ \
    "v_mad_u64_u32 v[60:61], s[94:95], v68, " S4 ", v[82:83]\n"   /* x = xb + q.lo * mult */         \
    "v_add_u32 v61, v61, v64\n"

// The kernel (third generation: the loop above, unrolled four times, fed by a producer wave).  Expanding a batch into items
// (prefix sums, LDS writes, the table gather) and writing a batch's staged words to the stream took ~15 % of the coder's
// time between two serial walks when one wave did everything (second generation); here a second wavefront of the
// workgroup does both, a step ahead of / behind the coder, with double-buffered item and staging areas and one barrier per
// 64-symbol batch.  Same stream, bit for bit.
__global__ __launch_bounds__(128) void rans_encode_kernel3(const int32_t* __restrict__ sym, const int32_t* __restrict__ idx,
                                                          const int64_t* __restrict__ sym_base,
                                                          const int64_t* __restrict__ counts, int split, DevTables t0,
                                                          DevTables t1, uint32_t* __restrict__ out, int64_t cap_words,
                                                          int64_t* __restrict__ out_words, int* __restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char esm[];
    // LDS: 2 x items [ENC_ITEM_CAP + 2][8 dwords] | 2 x staged words [ENC_ITEM_CAP] | control words | row metadata [nrows] int3
    uint32_t* items2 = reinterpret_cast<uint32_t*>(esm);
    uint32_t* obuf2 = items2 + 2 * (ENC_ITEM_CAP + 2) * 8;
    volatile uint32_t* ctl = obuf2 + 2 * ENC_ITEM_CAP;  // [0..1] items of a batch, [2..3] words of a batch, [4] overflow
    int3* rowmeta = reinterpret_cast<int3*>(obuf2 + 2 * ENC_ITEM_CAP + 8);
    const int s = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;  // 0: the serial coder, 1: producer of its items and writer of its words
    const DevTables& t = s < split ? t0 : t1;
    const int64_t n = counts[s];
    const int64_t base = sym_base[s];
    uint32_t* o = out + (size_t)s * cap_words;
    uint64_t x = RANS_LOW;
    int64_t w = cap_words;  // next free slot is w-1
    int bad = 0;

    for (int i = threadIdx.x; i < t.nrows; i += 128) rowmeta[i] = make_int3(t.row_off[i], t.sizes[i], t.offsets[i]);
    if (threadIdx.x < 8) ctl[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t items_addr0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(unsigned char*)items2;
    const uint32_t obuf_top0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(unsigned char*)(obuf2 + ENC_ITEM_CAP);

    // stage 1 of a batch (two batches ahead of the serial loop): symbol and index from HBM
    // stage 2 (one batch ahead): row geometry, escape split, gather of the symbol's reciprocal entry
    struct Half {
        uint4 e;       // table entry of the (clamped) symbol
        uint32_t raw;  // escape payload
        int nn;        // payload nibbles (escape only)
        int c;         // items of this symbol: 0 (invalid lane), 1, or nn + 2
    };
    auto finish = [&](const EncRaw& r) -> Half {
        Half hf;
        hf.e = make_uint4(~0u, ~0u, 65535u, 1u);
        hf.raw = 0;
        hf.nn = 0;
        hf.c = 0;
        if (r.valid) {
            const int3 rm = rowmeta[r.ti];  // {row_off, cdf_length, offset}
            const int top = rm.y - 2;
            int v = r.sv - rm.z;
            uint32_t raw = 0;
            bool esc = false;
            if (v < 0) {
                raw = (uint32_t)(-2 * v - 1);
                v = top;
                esc = true;
            } else if (v >= top) {
                raw = (uint32_t)(2 * (v - top));
                v = top;
                esc = true;
            }
            hf.e = reinterpret_cast<const uint4*>(t.enc)[rm.x + v];
            hf.raw = raw;
            int nn = 0;
            while (nn < 8 && (raw >> (nn * ESC_BITS)) != 0) ++nn;
            hf.nn = nn;
            hf.c = esc ? nn + 2 : 1;
        }
        return hf;
    };
    // items of a batch into LDS in WALK order (position 0 is coded first): the last symbol of the batch first, and per
    // escape symbol the payload nibbles from the most significant one, the count, then the escape slot itself (the reverse
    // of the decoder's order, rans_interface.cpp:147-162); padded to a multiple of four with no-op items.  Returns the count.
    auto expand = [&](const Half& hf, uint32_t* items) -> int {
        int incl = hf.c;  // inclusive prefix sum over the lanes
        int total;
        if (__builtin_amdgcn_ballot_w64(hf.c > 1) == 0) {
            // no escape in this batch (the common case at a trained model's rates): one item per valid lane, and the
            // valid lanes are a prefix of the wave
            total = __builtin_popcountll(__builtin_amdgcn_ballot_w64(hf.c != 0));
            incl = lane + 1;
        } else {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d, 64);
                if (lane >= d) incl += up;
            }
            total = __builtin_amdgcn_readlane(incl, 63);
        }
        if (hf.c) {
            const int o0 = incl - hf.c;  // first item index of this symbol; walk position = total - 1 - index
            auto put = [&](int index, uint32_t mlo, uint32_t mhi, uint32_t bias, uint32_t shift, uint32_t mult, uint32_t thr) {
                uint4* p = reinterpret_cast<uint4*>(items + (size_t)(total - 1 - index) * 8);
                p[0] = make_uint4(mlo, mhi, bias, shift);
                p[1] = make_uint4(mult, thr, 0u, 0u);
            };
            put(o0, hf.e.x, hf.e.y, hf.e.z & 0x1FFFFu, hf.e.z >> 17, 65536u - hf.e.w, hf.e.w << 15);
            if (hf.c > 1) {
                put(o0 + 1, ~0u, ~0u, (uint32_t)hf.nn + 15u, 0u, 15u, 4096u << 15);
                for (int k = 0; k < hf.nn; ++k)
                    put(o0 + 2 + k, ~0u, ~0u, ((hf.raw >> (k * ESC_BITS)) & ESC_MAX) + 15u, 0u, 15u, 4096u << 15);
            }
        }
        const int padded = (total + 3) & ~3;  // the walk is unrolled four times
        if (lane < padded - total) {  // no-op items: never renormalise (thr > any x >> 47), x + 0 + q * 0
            uint4* p = reinterpret_cast<uint4*>(items + (size_t)(total + lane) * 8);
            p[0] = make_uint4(~0u, ~0u, 0u, 0u);
            p[1] = make_uint4(0u, 65536u << 15, 0u, 0u);
        }
        return padded;
    };

    // Step st (nb-1 ... -2), one barrier each: the producer expands batch st into items[st & 1] and writes batch st+2's
    // staged words to the stream; the coder walks batch st+1's items (expanded one step ago) into obuf[(st+1) & 1].
    const int64_t nb = (n + 63) >> 6;
    Half cur;
    EncRaw raw1;
    if (wave == 1) {
        cur = finish(enc_load(sym, idx, base + (nb - 1) * 64 + lane, nb > 0 && (nb - 1) * 64 + lane < n));
        raw1 = enc_load(sym, idx, base + (nb - 2) * 64 + lane, nb > 1);
    }
    for (int64_t st = nb - 1; st >= -2; --st) {
        if (wave == 1) {
            const bool stop = rfl(ctl[4]) != 0u;  // (the coder ran out of room: nothing more to prepare or write)
            if (st >= 0 && !stop) {
                const int nitems = expand(cur, items2 + (size_t)(st & 1) * (ENC_ITEM_CAP + 2) * 8);
                if (lane == 0) ctl[st & 1] = (uint32_t)nitems;
                cur = finish(raw1);                                             // batch st-1: gather in flight for a whole step
                raw1 = enc_load(sym, idx, base + (st - 2) * 64 + lane, st > 1);  // batch st-2
            }
            if (st + 2 <= nb - 1 && !stop) {
                // staged words of batch st+2 -> the stream, same relative order (the stream grows downwards as well)
                const uint32_t* obuf = obuf2 + (size_t)((st + 2) & 1) * ENC_ITEM_CAP;
                const int nout = (int)rfl(ctl[2 + ((st + 2) & 1)]);
                for (int i = lane; i < nout; i += 64) o[w - nout + i] = obuf[ENC_ITEM_CAP - nout + i];
                w -= nout;
            }
        } else if (st + 1 >= 0 && st + 1 <= nb - 1 && !bad) {
            const int par = (int)((st + 1) & 1);
            int nitems = (int)ctl[par];
            if (w < ENC_ITEM_CAP + 8) {  // worst case for one batch: every item a word (rgbd_rans_max_bytes sizes for this)
                bad = 1;
                if (lane == 0) ctl[4] = 1;
                nitems = 0;
            }
            const uint32_t obuf_top = obuf_top0 + (uint32_t)par * ENC_ITEM_CAP * 4u;
            uint32_t wa = obuf_top - 4u;  // LDS byte address of the next staged word (the staging area fills downwards)
            if (nitems > 0) {
                uint32_t ia = items_addr0 + (uint32_t)par * (ENC_ITEM_CAP + 2) * 32u;
                nitems = (int)rfl((uint32_t)nitems);
                asm volatile(
                    "s_mov_b64 s[92:93], exec\n"
                    "s_mov_b64 exec, 1\n"                     // uniform work: one lane
                    "v_mov_b64 v[60:61], %[x]\n"
                    "v_mov_b32 v84, 0\n"
                    "v_mov_b32 v85, 4\n"
                    "v_mov_b32 v87, 0\n"                      // high halves of the 64-bit addend pairs
                    "v_mov_b32 v89, 0\n"
                    "v_mov_b32 v91, 0\n"
                    "ds_read_b128 v[70:73], %[ia]\n"
                    "ds_read_b64 v[74:75], %[ia] offset:16\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "1:\n"
                    "ds_read_b128 v[76:79], %[ia] offset:32\n"   // next item (set b)
                    "ds_read_b64 v[80:81], %[ia] offset:48\n"
                    ENC_ITEM("v70", "v71", "v72", "v73", "v74", "v75")
                    "s_waitcnt lgkmcnt(1)\n"                     // set b has arrived (the staged word may still be in flight)
                    "ds_read_b128 v[70:73], %[ia] offset:64\n"   // the item after (set a; reads past the end hit the padding)
                    "ds_read_b64 v[74:75], %[ia] offset:80\n"
                    ENC_ITEM("v76", "v77", "v78", "v79", "v80", "v81")
                    "s_waitcnt lgkmcnt(1)\n"
                    "ds_read_b128 v[76:79], %[ia] offset:96\n"
                    "ds_read_b64 v[80:81], %[ia] offset:112\n"
                    ENC_ITEM("v70", "v71", "v72", "v73", "v74", "v75")
                    "s_waitcnt lgkmcnt(1)\n"
                    "ds_read_b128 v[70:73], %[ia] offset:128\n"
                    "ds_read_b64 v[74:75], %[ia] offset:144\n"
                    ENC_ITEM("v76", "v77", "v78", "v79", "v80", "v81")
                    "v_add_u32 %[ia], 0x80, %[ia]\n"
                    "s_sub_u32 %[n], %[n], 4\n"
                    "s_cmp_lg_u32 %[n], 0\n"
                    "s_waitcnt lgkmcnt(1)\n"
                    "s_cbranch_scc1 1b\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "v_mov_b64 %[x], v[60:61]\n"
                    "s_mov_b64 exec, s[92:93]\n"
                    : [x] "+v"(x), [ia] "+v"(ia), [wa] "+v"(wa), [n] "+s"(nitems)
                    :
                    : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74",
                      "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",
                      "v90", "v91", "s92", "s93", "s94", "s95", "vcc", "scc", "memory");
                x = rfl64(x);
                wa = rfl(wa);
            }
            const int nout = (int)((obuf_top - 4u - wa) >> 2);
            if (lane == 0) ctl[2 + par] = (uint32_t)nout;
            w -= nout;  // (the coder's own count: the overflow test above and the place of the two state words)
        }
        __syncthreads();
    }
    bad = (int)ctl[4];
    if (wave == 1) return;
    if (bad) {
        if (lane == 0) {
            *err = 1;
            out_words[s] = 0;
        }
        return;
    }
    if (lane == 0) {  // rans64.h:96-103: the two state words go in front
        o[w - 1] = (uint32_t)(x >> 32);
        o[w - 2] = (uint32_t)x;
        out_words[s] = cap_words - (w - 2);
    }
}

int launch_rans_encode(const int32_t* sym, const int32_t* idx, const int64_t* sym_base, const int64_t* counts,
                       int nstreams, int split, DevTables t0, DevTables t1, uint32_t* out, int64_t cap_words,
                       int64_t* out_words, int* err, hipStream_t s)
{
    if (nstreams <= 0) return RGBD_OK;
    if (cap_words % 64 || cap_words >= ((int64_t)1 << 31)) return RGBD_EINVAL;
    const size_t lds = (size_t)(t0.nrows > t1.nrows ? t0.nrows : t1.nrows) * sizeof(int3);
    static const bool v1 = getenv("RGBD_CODER_V1") != nullptr;  // A/B switch: the first-generation loop
    const size_t lds3 = 2 * ((size_t)(ENC_ITEM_CAP + 2) * 32 + (size_t)ENC_ITEM_CAP * 4) + 32 + lds;
    if (!v1 && lds3 <= 64 * 1024) {
        hipLaunchKernelGGL(rans_encode_kernel3, dim3(nstreams), dim3(128), lds3, s, sym, idx, sym_base, counts, split, t0, t1,
                           out, cap_words, out_words, err);
        HIP_TRY(hipGetLastError());
        return RGBD_OK;
    }
    if (lds > 60 * 1024) return RGBD_ENOSPC;
    hipLaunchKernelGGL(rans_encode_kernel, dim3(nstreams), dim3(64), lds, s, sym, idx, sym_base, counts, split, t0, t1, out,
                       cap_words, out_words, err);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

// The decoder loops in ISA, cut into pieces so that the two loops below (batches of narrow rows only / batches that contain
// rows wider than the 64 lanes) share them.  See rans_decode_kernel for what each piece does and why it is written this way.
#define RANS_DEC_PROLOGUE \
    "s_mov_b64 s[84:85], %[x]\n" \
    "s_mov_b32 %[m0s], m0\n" \
    "s_sub_u32 m0, %[j], 64\n" /* lane - 64 (lane selects use the low six bits) */ \
    "v_readlane_b32 %[lb], %[pkrow], m0\n" \
    "v_add_u32 v60, %[lb], %[lane4]\n" \
    "ds_read_b32 v58, v60\n"

#define RANS_DEC_BODY \
    "s_waitcnt lgkmcnt(0)\n" /* v58: this symbol's row slots (loaded one symbol ahead) */ \
    "s_andn2_b32 %[cum], 0xffff, s84\n" /* cumc = 0xFFFF - cum */ \
    "v_cmp_lt_u16 vcc, %[cum], v58\n" /* low halves: cumc < 0x10000 - cdf[i+1]  <=>  cdf[i+1] <= cum */ \
    "s_lshr_b64 s[86:87], s[84:85], 16\n" /* x >> 16 for the update below */ \
    "s_bcnt1_i32_b64 %[a], vcc\n" /* = symbol index, <= 63 (a low half of 0 never compares) */ \
    "v_readlane_b32 %[t1], v58, %[a]\n" /* {0xFFFF - cdf[a] << 16 | 0x10000 - cdf[a+1]} */ \
    /* the row slots are dead from here on: the next symbol's row goes straight into v58 (lane 63's */ \
    /* successor is lane 0: a valid row, value unused), its LDS hop overlaps the state update below */ \
    "v_readlane_b32 %[lb], %[pkn], m0\n" \
    "v_readlane_b32 %[wn], %[wcur], %[wi]\n" /* next unread word (lane 64 wraps: never used then) */ \
    "v_writelane_b32 %[outv], %[a], m0\n" /* (an escape's value is written over it) */ \
    "v_add_u32 v60, %[lb], %[lane4]\n" \
    "ds_read_b32 v58, v60\n" \
    "s_lshr_b32 %[start], %[t1], 16\n" \
    "s_and_b32 %[e0], %[t1], 0xffff\n" /* 0: not resolved here (escape slot / wide row) */ \
    "s_min_u32 %[lb], %[e0], 1\n" /* ... as 0 / 1 for the loop test */ \
    "s_sub_u32 %[freq], %[start], %[e0]\n" \
    "s_add_u32 %[freq], %[freq], 1\n" \
    "s_sub_u32 %[t0], %[start], %[cum]\n" /* cum - cdf[a] */ \
    "s_mul_i32 %[t1], s87, %[freq]\n" /* x = freq * (x >> 16) + (cum - start) */ \
    "s_mul_hi_u32 s85, s86, %[freq]\n" \
    "s_mul_i32 s84, s86, %[freq]\n" \
    "s_add_u32 %[t1], %[t1], s85\n" \
    "s_add_u32 s84, s84, %[t0]\n" \
    "s_addc_u32 s85, %[t1], 0\n" \
    "s_lshr_b64 s[86:87], s[84:85], 31\n" /* SCC = (x >= 2^31): keep; else x = x << 32 | next word */ \
    "s_cselect_b32 s85, s85, s84\n" \
    "s_cselect_b32 s84, s84, %[wn]\n" \
    "s_subb_u32 %[wi], %[wi], -1\n" /* consumed (SCC = 0): advance */ \
    "s_add_u32 m0, m0, 1\n" /* carries out of lane 63: batch done */ \
    "s_subb_u32 %[lb], %[lb], 1\n" /* borrows (0 - 1, or 1 - 1 - carry) when either holds: leave */ \
    "s_cbranch_scc0 1b\n"

#define RANS_DEC_DISPATCH63 \
    /* the loop has been left behind symbol m0 - 1: batch done, escape slot, or slot 63 of a wider row */ \
    "63:\n" \
    "s_cmp_lg_u32 %[e0], 0\n" \
    "s_cbranch_scc1 3f\n" \
    "s_cmp_eq_u32 %[start], 0xffff\n" \
    "s_cbranch_scc1 60f\n"

#define RANS_DEC_ESCAPE \
    /* escape (rans_interface.cpp:323-345): a = the escape slot; its table step is done.  The nibbles are */ \
    /* worked on a copy of the state (s[88:89], word index in lb, %[cum] = the next unread word) that is */ \
    /* committed at the end; a count nibble of 15 (more than 8 payload nibbles follow) or a nearly used-up */ \
    /* word window leaves to the C++ path with the committed state untouched.  Straight-line: every */ \
    /* renormalisation is a pair of selects on the SCC of the s_lshr_b64 that tests it.  The nn <= 8 payload */ \
    /* nibbles are taken in two steps at most, because the state can run dry only once in between: after */ \
    /* k = (bits(x) - 28) >> 2 nibbles it is below 2^31 and takes in a word w, and the other nn - k <= 7 */ \
    /* nibbles are then w's low bits, which cannot bring it (>= 2^59 after the word) below 2^31 again. */ \
    /* Step A takes min(k, nn) nibbles and renormalises if needed, step B the rest. */ \
    "64:\n" \
    "s_cmp_gt_u32 %[wi], 56\n" \
    "s_cbranch_scc1 9f\n" \
    "v_readlane_b32 %[cum], %[wcur], %[wi]\n" \
    "s_mov_b64 s[88:89], s[84:85]\n" \
    "s_and_b32 %[e0], s88, 15\n" /* count nibble nn */ \
    "s_lshr_b64 s[88:89], s[88:89], 4\n" \
    "s_lshr_b64 s[86:87], s[88:89], 31\n" \
    "s_cselect_b32 s89, s89, s88\n" \
    "s_cselect_b32 s88, s88, %[cum]\n" \
    "s_subb_u32 %[lb], %[wi], -1\n" \
    "s_cmp_gt_u32 %[e0], 8\n" \
    "s_cbranch_scc1 9f\n" /* a longer count: C++ path */ \
    "v_readlane_b32 %[cum], %[wcur], %[lb]\n" \
    "s_flbit_i32_b64 %[t0], s[88:89]\n" /* leading zeros (<= 32) */ \
    "s_sub_u32 %[t0], 36, %[t0]\n" \
    "s_lshr_b32 %[t0], %[t0], 2\n" /* k */ \
    "s_min_u32 %[t0], %[t0], %[e0]\n" /* step A: k' = min(k, nn) nibbles */ \
    "s_lshl_b32 %[t1], %[t0], 2\n" \
    "s_bfm_b64 s[86:87], %[t1], 0\n" \
    "s_and_b32 %[start], s88, s86\n" /* raw, low part */ \
    "s_lshr_b64 s[88:89], s[88:89], %[t1]\n" \
    "s_lshr_b64 s[86:87], s[88:89], 31\n" \
    "s_cselect_b32 s89, s89, s88\n" \
    "s_cselect_b32 s88, s88, %[cum]\n" \
    "s_subb_u32 %[lb], %[lb], -1\n" \
    "s_sub_u32 %[e0], %[e0], %[t0]\n" /* step B: the other nn - k' (<= 7; 0 unless A renormalised) */ \
    "s_lshl_b32 %[e0], %[e0], 2\n" \
    "s_bfm_b32 %[t0], %[e0], 0\n" \
    "s_and_b32 %[t0], s88, %[t0]\n" \
    "s_lshr_b64 s[88:89], s[88:89], %[e0]\n" \
    "s_lshl_b32 %[t0], %[t0], %[t1]\n" /* (k' = 8: this part is 0 and so is the 5-bit shift count) */ \
    "s_or_b32 %[start], %[start], %[t0]\n" /* raw */ \
    "s_lshr_b32 %[t1], %[start], 1\n" /* value: even raw -> last + raw / 2, odd -> -(raw >> 1) - 1 */ \
    "s_add_u32 %[t0], %[t1], %[a]\n" \
    "s_not_b32 %[t1], %[t1]\n" \
    "s_bitcmp1_b32 %[start], 0\n" \
    "s_cselect_b32 %[t1], %[t1], %[t0]\n" \
    "s_sub_u32 m0, m0, 1\n" /* (one scalar operand besides m0 is all v_writelane takes) */ \
    "v_writelane_b32 %[outv], %[t1], m0\n" \
    "s_add_u32 m0, m0, 1\n" \
    "s_mov_b64 s[84:85], s[88:89]\n" /* commit */ \
    "s_mov_b32 %[wi], %[lb]\n" \
    "s_sub_u32 %[t0], %[wi], m0\n" /* + symbols left in the batch: do they fit the word window? */ \
    "s_cmp_gt_u32 %[t0], 64\n" \
    "s_cbranch_scc1 92f\n" /* no: hand the realign to the caller */ \
    /* on with the next symbol (its row has been prefetched) */ \
    "65:\n" \
    "s_cmp_eq_u32 m0, 0\n" \
    "s_cbranch_scc1 3f\n" \
    "s_branch 1b\n"

#define RANS_DEC_WIDE60 \
    /* rows wider than the 64 lanes: the bucket table (cum >> shift -> first candidate, its start, its */ \
    /* frequency) and, behind it, a probe of the next 64 row entries.  The state is untouched (identity step). */ \
    "60:\n" \
    "s_sub_u32 m0, m0, 1\n" \
    "s_and_b32 %[cum], s84, 0xffff\n" \
    "v_readlane_b32 %[lb], %[lutbase], m0\n" \
    "s_lshr_b32 %[t0], %[cum], %[shift]\n" \
    "s_lshl3_add_u32 %[lb], %[t0], %[lb]\n" \
    "v_mov_b32 v62, %[lb]\n" \
    "ds_read_b64 v[62:63], v62\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_readfirstlane_b32 %[e0], v62\n" \
    "v_readfirstlane_b32 %[freq], v63\n" \
    "s_lshr_b32 %[start], %[e0], 16\n" \
    "s_sub_u32 %[t0], %[cum], %[start]\n" \
    "s_and_b32 %[a], %[e0], 0xffff\n" \
    "s_mov_b32 %[e0], 1\n" /* (from here on: 1 = table symbol, 0 = escape slot) */ \
    "s_cmp_ge_u32 %[t0], %[freq]\n" \
    "s_cbranch_scc0 68f\n" \
    /* second level: the symbol lies behind the bucket's first candidate a.  The lanes probe the 64 row */ \
    /* entries after a at once; k = #(entry < cum) locates it.  k = 64 (further away) is left to the C++ */ \
    /* path below; k = 0 (escape marker) and a probe ending on the pad are the row's escape slot. */ \
    "s_cmp_eq_u32 %[freq], 0\n" /* escape marker: the bucket's first candidate is the row's */ \
    "s_cbranch_scc1 71f\n" /* last slot, so the symbol is that slot -- no probe needed */ \
    "v_readlane_b32 %[lb], %[rowbase], m0\n" \
    "s_lshl1_add_u32 %[lb], %[a], %[lb]\n" \
    "v_add_u32 v62, %[lb], %[lane2]\n" \
    "ds_read_u16 v63, v62 offset:2\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_cmp_gt_u32 vcc, %[cum], v63\n" \
    "s_bcnt1_i32_b64 %[t1], vcc\n" \
    "s_cmp_eq_u32 %[t1], 64\n" \
    "s_cbranch_scc1 8f\n" \
    "s_cmp_eq_u32 %[t1], 0\n" \
    "s_cbranch_scc1 71f\n" /* escape marker: symbol a is the escape slot */ \
    "s_sub_u32 %[t0], %[t1], 1\n" \
    "v_readlane_b32 %[lb], v63, %[t1]\n" /* cm[a + 1 + k]  = next start - 1 */ \
    "v_readlane_b32 %[t0], v63, %[t0]\n" /* cm[a + k]      = start - 1 */ \
    "s_cmp_eq_u32 %[lb], 0xffff\n" \
    "s_cbranch_scc1 72f\n" /* the row's last slot: escape */ \
    "s_add_u32 %[a], %[a], %[t1]\n" \
    "s_sub_u32 %[freq], %[lb], %[t0]\n" \
    "s_sub_u32 %[t0], %[cum], %[t0]\n" \
    "s_sub_u32 %[t0], %[t0], 1\n" /* cum - start */ \
    "s_branch 68f\n" \
    "71:\n" \
    "s_sub_u32 %[t0], %[cum], %[start]\n" \
    "s_sub_u32 %[freq], 0x10000, %[start]\n" \
    "s_mov_b32 %[e0], 0\n" \
    "s_branch 68f\n" \
    "72:\n" \
    "s_add_u32 %[a], %[a], %[t1]\n" \
    "s_sub_u32 %[freq], 0xffff, %[t0]\n" \
    "s_sub_u32 %[t0], %[cum], %[t0]\n" \
    "s_sub_u32 %[t0], %[t0], 1\n" \
    "s_mov_b32 %[e0], 0\n" \
    "68:\n" /* a, freq, cum - start: the table step, as in the loop */ \
    "v_writelane_b32 %[outv], %[a], m0\n" \
    "s_lshr_b64 s[86:87], s[84:85], 16\n" \
    "s_mul_i32 %[t1], s87, %[freq]\n" \
    "s_mul_hi_u32 s85, s86, %[freq]\n" \
    "s_mul_i32 s84, s86, %[freq]\n" \
    "s_add_u32 %[t1], %[t1], s85\n" \
    "s_add_u32 s84, s84, %[t0]\n" \
    "s_addc_u32 s85, %[t1], 0\n" \
    "s_lshr_b64 s[86:87], s[84:85], 31\n" \
    "s_cselect_b32 s85, s85, s84\n" \
    "s_cselect_b32 s84, s84, %[wn]\n" \
    "s_subb_u32 %[wi], %[wi], -1\n" \
    "s_add_u32 m0, m0, 1\n" \
    "s_cmp_eq_u32 %[e0], 0\n" \
    "s_cbranch_scc1 64b\n" /* escape slot: the nibbles */ \
    "s_branch 65b\n"

// Rows of 129 ... 4032 slots in the loop for batches with several symbols on such rows (round 4).  The lanes of such a batch
// point at the rows' COARSE first level (build_tables: slot j = the block of `stride` symbols from j * stride on), so the same 16-bit compare that resolves
// a narrow symbol yields the block, and ONE more hop -- the 64 entries of the cdf - 1 array behind the block's first --
// yields the symbol: k = #(entry < cum), start = k ? entry[k - 1] + 1 : the block's start (high half of the coarse slot),
// end = entry[k] + 1 (the pad 0xFFFF behind a row makes that 65536 for the last, i.e. escape, slot).  The next symbol's first
// level is requested right behind that hop and waited for at the next loop top; the table step and the loop edge (escape flag
// and batch end in one s_subb) are those of the narrow body.
#define RANS_DEC_WIDE20 \
    "20:\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "s_andn2_b32 %[cum], 0xffff, s84\n" \
    "v_cmp_lt_u16 vcc, %[cum], v58\n" \
    "s_bcnt1_i32_b64 %[a], vcc\n"                /* block */ \
    "v_readlane_b32 %[lb], %[rowbase], m0\n" \
    "v_readlane_b32 %[t0], %[wstep], m0\n"       /* 2 * stride */ \
    "s_mul_i32 %[e0], %[a], %[t0]\n" \
    "s_add_u32 %[lb], %[lb], %[e0]\n" \
    "v_add_u32 v62, %[lb], %[lane2]\n" \
    "ds_read_u16 v63, v62 offset:2\n"            /* cdf[block * stride + 1 + lane] - 1 */ \
    "v_readlane_b32 %[t1], v58, %[a]\n"          /* {cdf[block * stride] << 16 | ...} */ \
    "v_readlane_b32 %[lb], %[pkn], m0\n" \
    "v_readlane_b32 %[wn], %[wcur], %[wi]\n" \
    "v_add_u32 v60, %[lb], %[lane4]\n" \
    "ds_read_b32 v58, v60\n"                     /* the next symbol's first level */ \
    "s_lshr_b32 s88, %[e0], 1\n"                 /* block * stride */ \
    "s_lshr_b32 %[start], %[t1], 16\n" \
    "s_and_b32 %[cum], s84, 0xffff\n" \
    "s_lshr_b64 s[86:87], s[84:85], 16\n" \
    "s_waitcnt lgkmcnt(1)\n" \
    "v_cmp_gt_u32 vcc, %[cum], v63\n" \
    "s_bcnt1_i32_b64 %[t1], vcc\n"               /* k */ \
    "s_add_u32 %[a], s88, %[t1]\n" \
    "v_writelane_b32 %[outv], %[a], m0\n" \
    "v_readlane_b32 %[lb], v63, %[t1]\n"         /* cdf[a + 1] - 1 */ \
    "s_sub_u32 %[t0], %[t1], 1\n" \
    "v_readlane_b32 %[t0], v63, %[t0]\n"         /* cdf[a] - 1 (k >= 1; k = 0 reads lane 63 and is deselected) */ \
    "s_add_u32 %[t0], %[t0], 1\n" \
    "s_cmp_eq_u32 %[t1], 0\n" \
    "s_cselect_b32 %[start], %[start], %[t0]\n" \
    "s_sub_u32 %[e0], 0xffff, %[lb]\n"           /* 0: the row's last slot (escape) */ \
    "s_sub_u32 %[freq], %[lb], %[start]\n" \
    "s_add_u32 %[freq], %[freq], 1\n" \
    "s_min_u32 %[lb], %[e0], 1\n" \
    "s_sub_u32 %[t0], %[cum], %[start]\n" \
    "s_mul_i32 %[t1], s87, %[freq]\n" \
    "s_mul_hi_u32 s85, s86, %[freq]\n" \
    "s_mul_i32 s84, s86, %[freq]\n" \
    "s_add_u32 %[t1], %[t1], s85\n" \
    "s_add_u32 s84, s84, %[t0]\n" \
    "s_addc_u32 s85, %[t1], 0\n" \
    "s_lshr_b64 s[86:87], s[84:85], 31\n" \
    "s_cselect_b32 s85, s85, s84\n" \
    "s_cselect_b32 s84, s84, %[wn]\n" \
    "s_subb_u32 %[wi], %[wi], -1\n" \
    "s_add_u32 m0, m0, 1\n" \
    "s_subb_u32 %[lb], %[lb], 1\n" \
    "s_cbranch_scc0 1b\n" \
    "s_cmp_lg_u32 %[e0], 0\n" \
    "s_cbranch_scc1 3f\n"                        /* batch done */ \
    "s_branch 64b\n"                             /* escape slot: the nibbles */

#define RANS_DEC_EXITS \
    "3:\n" \
    "s_mov_b32 %[more], 0\n" \
    "s_branch 5f\n" \
    "92:\n" \
    "s_mov_b32 %[more], 2\n" \
    "s_branch 5f\n" \
    "9:\n" \
    "s_mov_b32 %[more], 3\n" \
    "s_branch 5f\n" \
    "8:\n" \
    "s_mov_b32 %[more], 1\n" \
    "5:\n" \
    "s_waitcnt lgkmcnt(0)\n" /* drain the row prefetch */ \
    "s_mov_b64 %[x], s[84:85]\n" \
    "s_add_u32 %[j], m0, 64\n" \
    "s_mov_b32 m0, %[m0s]\n"

// ---------------------------------------------------------------------------------------------
// Decoder.  LDS holds the packed u16 CDF rows and a per-row bucket table (cum >> (16 - lut_bits) -> first candidate
// symbol, its start and its frequency: one 8-byte read resolves a symbol unless a boundary falls inside the bucket).
// 256 threads fill LDS, then wave 0 decodes.  The (x, pos) state persists in HBM between the 20 per-part launches of
// one stream (RansDecoder::decode_stream semantics).
__global__ __launch_bounds__(256) void rans_decode_kernel(const uint32_t* __restrict__ streams,
                                                          const int64_t* __restrict__ stream_off,
                                                          const int64_t* __restrict__ stream_len,
                                                          uint64_t* __restrict__ state, int init,
                                                          const int32_t* __restrict__ idx, int32_t* __restrict__ sym,
                                                          const int64_t* __restrict__ sym_base, int64_t part_off,
                                                          int64_t count, DevTables t, int nstreams, int spw)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int lut_n = (1 << t.lut_bits) + 1;
    uint2* sl = reinterpret_cast<uint2*>(dsm);                                          // [nrows][lut_n] bucket table
    uint2* rowinfo = sl + ((t.nrows * lut_n + 1) & ~1);                                  // [nrows] packed row info
    // cm: the packed rows again, each followed by 64 pad entries, holding cdf - 1 (entry 0: 0; pads: 0xFFFF = 65536 - 1).
    // "entry < cum" is then "cdf <= cum" for every lane of a 64-wide probe, pads included, and a probe that runs into
    // the pad has found the row's last (= escape) slot.  Row r starts at row_off[r] + 64 * r.
    uint16_t* cm = reinterpret_cast<uint16_t*>(rowinfo + ((t.nrows + 1) & ~1));           // [t.total + 64 * nrows]
    // pk: the first 64 slots of every row as {cdf[i] << 16 | cdf[i + 1] - 1} (build_tables): the hot loop's only table
    uint32_t* pk = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(cm) +
                                               (((size_t)(t.total + 64 * t.nrows) * 2 + 15) & ~(size_t)15));  // [nrows][64]
    uint32_t* pkc = pk + (size_t)t.nrows * 64;  // [ncoarse][64]: the coarse first level of the rows with 129 ... 4032 slots

    // Four streams per workgroup, one per wavefront (each on its own SIMD): the ~140 KB of tables in LDS are shared, so a
    // decode launch of 16 streams holds 4 CUs' LDS instead of 16 (a conv workgroup cannot co-reside with these tables).
    // (spw = streams per workgroup: 4, or 1 for a call with one or two streams -- a single image -- where sharing the
    // LDS between waves only costs latency)
    const int tid = threadIdx.x;
    const int s = blockIdx.x * spw + (tid >> 6);
    {
        const int ncm16 = ((t.total + 64 * t.nrows) * 2 + 15) / 16;  // both images are padded to 16 bytes
        const uint4* gcm = reinterpret_cast<const uint4*>(t.cm);
        uint4* lcm = reinterpret_cast<uint4*>(cm);
        for (int i = tid; i < ncm16; i += 256) lcm[i] = gcm[i];
        const uint4* gpk = reinterpret_cast<const uint4*>(t.pk);
        uint4* lpk = reinterpret_cast<uint4*>(pk);
        for (int i = tid; i < t.nrows * 16; i += 256) lpk[i] = gpk[i];
        const uint4* gpc = reinterpret_cast<const uint4*>(t.pkc);
        uint4* lpc = reinterpret_cast<uint4*>(pkc);
        for (int i = tid; i < t.ncoarse * 16; i += 256) lpc[i] = gpc[i];
        const uint2* gl = reinterpret_cast<const uint2*>(t.lut);  // bucket entries (escape candidates carry frequency 0)
        for (int i = tid; i < t.nrows * lut_n; i += 256) sl[i] = gl[i];
    }
    for (int i = tid; i < t.nrows; i += 256) {  // {row start in cm : 16 | cdf_length : 16}, {offset (signed) : 16 | wstep : 8 | coarse row : 8}
        // wstep: 0, or for a row with a coarse first level the byte step of its blocks in cm (2 * ceil(slots / 64))
        const int cr = t.coarse[i];
        const uint32_t wstep = cr >= 0 ? 2u * (uint32_t)((t.sizes[i] - 1 + 63) / 64) : 0u;
        rowinfo[i] = make_uint2((uint32_t)(t.row_off[i] + 64 * i) | ((uint32_t)t.sizes[i] << 16),
                                ((uint32_t)t.offsets[i] & 0xFFFFu) | (wstep << 16) | ((uint32_t)(cr >= 0 ? cr : 0) << 24));
    }
    __syncthreads();
    if ((tid >> 6) >= spw || s >= nstreams) return;
    const int lane = tid & 63;

    const uint32_t* st = streams + stream_off[s];
    const int64_t nwords = stream_len[s];
    uint64_t x;
    int64_t pos;
    if (init) {
        x = (uint64_t)(nwords > 0 ? st[0] : 0u) | ((uint64_t)(nwords > 1 ? st[1] : 0u) << 32);  // rans64.h:107-115
        pos = 2;
    } else {
        x = state[2 * s];
        pos = (int64_t)state[2 * s + 1];
    }
    x = rfl64(x);
    pos = (int64_t)rfl64((uint64_t)pos);
    const int64_t base = sym_base[s] + part_off;
    const int shift = 16 - t.lut_bits;

    // stream words: lane k of wcur holds word wpos0 + k, wnext the following 64 words (already in flight); wi = index
    // of the next unread word inside wcur.  The window is re-aligned (wi = 0) at the start of every 64-symbol batch and
    // behind an escape whose words leave fewer than one per remaining symbol of the batch, so the ordinary symbols (at most
    // one word each) can never run past lane 63 and the hot loop reads words with a bare readlane.
    int64_t wpos0 = pos;
    int wi = 0;
    uint32_t wcur = (wpos0 + lane < nwords) ? st[wpos0 + lane] : 0u;
    uint32_t wnext = (wpos0 + 64 + lane < nwords) ? st[wpos0 + 64 + lane] : 0u;
    auto realign = [&]() {
        if (wi == 0) return;
        const int src = ((lane + wi) & 63) << 2;
        const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)wcur);
        const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)wnext);
        wcur = (lane + wi < 64) ? a : c;
        wpos0 += wi;
        wi = 0;
        wnext = (wpos0 + 64 + lane < nwords) ? st[wpos0 + 64 + lane] : 0u;
    };
    auto next_word_checked = [&]() -> uint32_t {  // escape path only: may cross the window
        if (wi == 64) {
            wi = 0;
            wpos0 += 64;
            wcur = wnext;
            wnext = (wpos0 + 64 + lane < nwords) ? st[wpos0 + 64 + lane] : 0u;
        }
        return rdl(wcur, wi++);
    };
    const uint32_t dsm_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)dsm;
    const uint32_t cm_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(unsigned char*)cm;
    const uint32_t pk_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(unsigned char*)pk;
    const uint32_t pkc_addr = pk_addr + (uint32_t)t.nrows * 256u;
    const uint32_t lane2 = (uint32_t)lane * 2u, lane4 = (uint32_t)lane * 4u;

    // A batch is 64 symbols, one per lane; a shorter (last) batch sits in the TOP lanes (symbol k in lane k + 64 - cnt),
    // so that the loop counter always ends at lane 64: m0 = lane - 64 then wraps to 0 with a carry, and the increment is
    // the loop test.
    const int64_t nb = (count + 63) >> 6;
    auto load_ti = [&](int64_t bb) -> int {
        const int64_t rem = count - bb * 64;
        const int sh_ = rem < 64 ? 64 - (int)rem : 0;
        return lane >= sh_ ? idx[base + bb * 64 + lane - sh_] : 0;
    };
    int ti_next = load_ti(0);
    bool pend_ok = false;  // symbols of the batch before, not yet stored
    int64_t pend_pos = 0;
    int32_t pend_val = 0;
    for (int64_t b = 0; b < nb; ++b) {
        const int cnt = (int)((count - b * 64) < 64 ? (count - b * 64) : 64);
        const int sh = 64 - cnt;
        const int ti = ti_next;
        const uint32_t lutbase = dsm_addr + (uint32_t)(ti * lut_n) * 8u;  // LDS address of this lane's symbol's bucket row
        const uint2 rinfo = rowinfo[ti];  // this lane's row: {start : 16 | cdf_length : 16}, {offset : 16}
        const uint32_t rowbase = cm_addr + (rinfo.x & 0xFFFFu) * 2u;  // LDS address of the row in cm
        // Rows with a coarse first level (129 ... 4032 slots): a batch with at least five symbols on such rows runs the loop
        // that resolves them in two hops (~156 ns each instead of ~286 through the bucket table, at ~11 ns per narrow symbol of
        // the batch for the test in front: the break-even is five); its lanes then point at the coarse slots.  Any other
        // batch runs the plain loop, every wide row through the bucket table.
        const uint32_t wstep = (rinfo.y >> 16) & 0xFFu;
        uint64_t wmask = __builtin_amdgcn_ballot_w64(wstep != 0u && lane >= sh);
        if (__builtin_popcountll(wmask) < 5) wmask = 0;
        const bool coarse_lane = (wmask >> lane) & 1u;
        const uint32_t pkrow = coarse_lane ? pkc_addr + (rinfo.y >> 24) * 256u : pk_addr + (uint32_t)ti * 256u;  // first-level slots
        // the same for the NEXT lane's symbol: the loop prefetches symbol m0 + 1's slots with lane select m0
        const uint32_t pkn = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, (int)pkrow);
        uint32_t outv = 0;
        int j = sh;
        realign();  // before the prefetch below: its wait then only covers loads issued a whole batch ago
        if (b + 1 < nb) ti_next = load_ti(b + 1);  // prefetch
        // the previous batch's symbols leave HERE, in front of a whole serial walk: the wait at the top of the next batch
        // (for the index prefetch above, in issue order) then finds the store long done instead of just issued (~0.5 us
        // per batch when it sat at the end of the loop body)
        if (pend_ok) sym[pend_pos] = pend_val;
        while (j < 64) {
            // hot loop: symbols the first level resolves.  The word window is loop-invariant here; anything else (a row
            // wider than the lanes, an unusual escape) leaves the loop, is finished below and the loop is re-entered
            // behind it.
            // Written in ISA.  The loop is one dependent chain (x -> cum -> slot -> x) run by a lone wave, and what it
            // costs was measured one piece at a time (knock-out builds, tools/ko_probe.py): ~2 ns of issue per scalar
            // instruction, 3-6 per vector one, ~9 per branch, taken or not.  So: 31 instructions and ONE branch per symbol
            // -- the loop edge, whose condition also carries "this was an escape slot": the escape slot and slot 63 of a
            // wider row take the ordinary update too (for the latter the slot encodes the identity step, freq = 65536,
            // start = 0) and are told apart after the loop has been left.  The slot word holds complements so that
            // cumc = 0xFFFF & ~x is one instruction and the s_and that splits the word yields "not resolved" as 0;
            // s_lshr_b64 sets SCC = (result != 0), which IS the "no renormalisation" flag; the loop counter's increment
            // (lane - 64, so it carries out of lane 63) and the escape flag meet in one s_subb.  (v_cmpx + v_readfirstlane
            // instead of v_cmp / s_bcnt1 / v_readlane was built too: two more instructions and a wait -- the lane read
            // sees the old EXEC for 4 cycles, tools/ubench/cmpx_first.hip -- for no gain.)
            //   s[84:85] = x, s[86:87] = scratch pair, s[88:89] = the escape block's copy of x, v58 = slots, v60 = address
            uint32_t a, start, freq, cum, more;
            {
                uint32_t lb, t0, t1, e0, m0s, wn;
                // Two copies of the loop in one statement.  wmask == 0: the plain loop (second copy; wide rows leave it through
                // the identity slot for the bucket-table path at 60:).  Otherwise the first copy: the same loop with one test
                // per symbol in front (bit m0 of wmask: +2 instructions and a branch) and the two-level body at 20: for the
                // symbols on rows with a coarse first level.
                asm volatile(
                    "s_cmp_eq_u64 %[wmask], 0\n"
                    "s_cbranch_scc1 90f\n"
                    RANS_DEC_PROLOGUE
                    ".p2align 6\n"
                    "1:\n"
                    "s_bitcmp1_b64 %[wmask], m0\n"
                    "s_cbranch_scc1 20f\n"
                    RANS_DEC_BODY
                    RANS_DEC_DISPATCH63
                    RANS_DEC_ESCAPE
                    RANS_DEC_WIDE20
                    RANS_DEC_WIDE60
                    RANS_DEC_EXITS
                    "s_branch 99f\n"
                    "90:\n"
                    RANS_DEC_PROLOGUE
                    ".p2align 6\n"
                    "1:\n"
                    RANS_DEC_BODY
                    RANS_DEC_DISPATCH63
                    RANS_DEC_ESCAPE
                    RANS_DEC_WIDE60
                    RANS_DEC_EXITS
                    "99:\n"
                    : [x] "+s"(x), [j] "+s"(j), [wi] "+s"(wi), [outv] "+v"(outv), [a] "=&s"(a), [start] "=&s"(start),
                      [freq] "=&s"(freq), [cum] "=&s"(cum), [more] "=&s"(more), [lb] "=&s"(lb), [t0] "=&s"(t0),
                      [t1] "=&s"(t1), [e0] "=&s"(e0), [m0s] "=&s"(m0s), [wn] "=&s"(wn)
                    : [shift] "s"(shift), [lutbase] "v"(lutbase), [wcur] "v"(wcur), [rowbase] "v"(rowbase),
                      [lane2] "v"(lane2), [pkrow] "v"(pkrow), [pkn] "v"(pkn), [lane4] "v"(lane4), [wstep] "v"(wstep),
                      [wmask] "s"(wmask)
                    : "s84", "s85", "s86", "s87", "s88", "s89", "v58", "v59", "v60", "v62", "v63", "vcc", "scc", "memory");
            }
            more = rfl(more);
            if (!more) break;
            if (more == 2) {  // an escape used up the word window: symbol j-1 is done, re-align before going on
                realign();
                continue;
            }
            // rare shapes: generic path.  more == 1: symbol j from the bucket entry (candidate more than 64 entries away);
            // more == 3: symbol j - 1 was an escape slot whose table step is done -- only its nibbles are left (a count
            // above 8, or the word window nearly used up)
            const int jj = more == 3 ? j - 1 : j;
            const uint32_t r0 = rdl(rinfo.x, jj);
            const int ro = (int)(r0 & 0xFFFFu), last = (int)(r0 >> 16) - 2;  // last = escape slot
            if (more == 1) {
                cum = (uint32_t)x & 0xFFFFu;
                {
                    const uint2 ev = sl[rdl((uint32_t)ti, j) * lut_n + (int)(cum >> shift)];
                    const uint32_t e0 = rfl(ev.x);
                    a = e0 & 0xFFFFu;
                    start = e0 >> 16;
                    freq = rfl(ev.y);
                }
                if ((int)a == last) freq = 65536u - start;                        // undo the fast-path marker
                if (cum - start >= freq) {
                    // the symbol lies behind the bucket's first candidate: the 64 lanes compare the next 64 row entries
                    // with cum at once (one LDS read + ballot per 64 candidates instead of one dependent read per candidate)
                    uint32_t lo = a + 1, prev = start;
                    for (;;) {
                        const int ci = (int)lo + lane;
                        const uint32_t c = (ci <= last) ? (uint32_t)cm[ro + ci] + 1u : 65536u;  // entry i = start of symbol i
                        const int k = __builtin_popcountll(__builtin_amdgcn_ballot_w64(c <= cum));
                        if (k == 64) {
                            prev = rdl(c, 63);
                            lo += 64;
                            continue;
                        }
                        a = lo + (uint32_t)k - 1u;
                        start = k ? rdl(c, k - 1) : prev;
                        freq = rdl(c, k) - start;
                        break;
                    }
                }
                x = (uint64_t)freq * (x >> PROB_BITS) + (cum - start);
                if ((uint32_t)(x >> 31) == 0u) x = (x << 32) | rdl(wcur, wi++);
                ++j;
            } else {
                a = (uint32_t)last;
            }
            int v = (int)a;
            if ((int)a == last) {  // escape: rans_interface.cpp:323-345 (4-bit nibbles, 80-96)
                auto bits = [&]() -> int {
                    const int val = (int)((uint32_t)x & ESC_MAX);
                    x >>= ESC_BITS;
                    if ((uint32_t)(x >> 31) == 0u) x = (x << 32) | next_word_checked();
                    return val;
                };
                int nib = bits();
                int nn = nib;
                while (nib == (int)ESC_MAX) {
                    nib = bits();
                    nn += nib;
                }
                int raw = 0;
                for (int k = 0; k < nn; ++k) {
                    nib = bits();
                    if (k < 8) raw |= nib << (k * ESC_BITS);
                }
                v = raw >> 1;
                if (raw & 1) v = -v - 1;
                else v += last;
                // an escape may have used many words: restore the window invariant -- but only when the rest of the
                // batch (one word per symbol at most) could run past lane 63; a realign waits for a fresh global load
                if (wi + (64 - j) > 64) realign();
            }
            outv = wrl((uint32_t)v, jj, outv);
        }
        // table offset of each symbol (entropy_models' _offset) added by its own lane, then one coalesced store
        const int off = (int)(int16_t)(rinfo.y & 0xFFFFu);
        pend_ok = lane >= sh;
        pend_pos = base + b * 64 + lane - sh;
        pend_val = (int32_t)outv + off;
    }
    if (pend_ok) sym[pend_pos] = pend_val;
    if (lane == 0) {
        state[2 * s] = x;
        state[2 * s + 1] = (uint64_t)(wpos0 + wi);
    }
}

size_t rans_decode_lds_bytes(const DevTables& t)
{
    const size_t lut_n = ((size_t)1 << t.lut_bits) + 1;
    return (((size_t)t.nrows * lut_n + 1) & ~(size_t)1) * 8 + (((size_t)t.nrows + 1) & ~(size_t)1) * 8 +
           ((((size_t)t.total + 64 * (size_t)t.nrows) * 2 + 15) & ~(size_t)15) + ((size_t)t.nrows + (size_t)t.ncoarse) * 256;
}

int launch_rans_decode(const uint32_t* streams, const int64_t* stream_off_words, const int64_t* stream_len_words,
                       int nstreams, uint64_t* state, int init, const int32_t* idx, int32_t* sym,
                       const int64_t* sym_base, int64_t part_off, int64_t count, DevTables t, hipStream_t s)
{
    if (nstreams <= 0 || count <= 0) return RGBD_OK;
    const size_t lds = rans_decode_lds_bytes(t);
    if (lds > 158 * 1024 || t.nrows * (((size_t)1 << t.lut_bits) + 1) > 65535 || t.total + 64 * t.nrows > 65535)
        return RGBD_ENOSPC;
    {   // the attribute is per device and this is called from several host threads (CodecPool)
        static std::mutex mu;
        static bool configured[64] = {false};
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !configured[dev]) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(rans_decode_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dev >= 0 && dev < 64) configured[dev] = true;
        }
    }
    const int spw = nstreams <= 2 ? 1 : 4;
    hipLaunchKernelGGL(rans_decode_kernel, dim3((nstreams + spw - 1) / spw), dim3(256), lds, s, streams, stream_off_words,
                       stream_len_words, state, init, idx, sym, sym_base, part_off, count, t, nstreams, spw);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}
