// The stand-alone entropy-coding operators of the C ABI (include/rgbd_amd.h): CDF construction, the packed device tables, the
// host-buffer rANS coder behind the reference's RansEncoder / RansDecoder (rans_interface.cpp:99-351), its device-batched
// form (symbols, indexes and streams stay in HBM; many streams per launch) and the checkerboard quantise / index step
// (utils/ckbd.py:83-125 + entropy_models.py:118-146,561-568).  The codec (engine.hip) calls the same kernels directly.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/rgbd_amd.h"
#include "engine_internal.h"

int build_tables(const int32_t* cdf, int stride, const int32_t* sizes, const int32_t* offsets, int nrows, TableSet* ts)
{
    if (!cdf || !sizes || !offsets || nrows <= 0 || stride < 3) return RGBD_EINVAL;
    std::vector<int32_t> row_off(nrows);
    int total = 0;
    for (int r = 0; r < nrows; ++r) {
        if (sizes[r] < 3 || sizes[r] > stride) return RGBD_EINVAL;
        row_off[r] = total;
        total += sizes[r] - 1;  // final 65536 entry is implicit
    }
    std::vector<uint16_t> packed(total);
    // bucket-table resolution: as fine as fits next to the packed rows in one CU's LDS (160 KiB)
    int bits = 8;
    while (bits > 4 &&
           (size_t)nrows * ((1u << bits) + 1) * 8 + ((size_t)total + 64 * (size_t)nrows) * 2 + (size_t)nrows * 256 > 148 * 1024)
        --bits;
    const int LN = (1 << bits) + 1;
    std::vector<uint32_t> lut((size_t)nrows * LN * 2);  // {j | row[j] << 16, freq_j}
    for (int r = 0; r < nrows; ++r) {
        const int32_t* row = cdf + (size_t)r * stride;
        const int len = sizes[r];
        if (row[0] != 0 || row[len - 1] != 65536) return RGBD_EINVAL;
        for (int j = 0; j < len - 1; ++j) {
            if (row[j] < 0 || row[j] > 65535 || row[j + 1] <= row[j]) return RGBD_EINVAL;
            packed[row_off[r] + j] = (uint16_t)row[j];
        }
        int j = 0;
        for (int b = 0; b < LN; ++b) {
            const int64_t lim = (int64_t)b << (16 - bits);
            while (j + 1 <= len - 2 && row[j + 1] <= lim) ++j;
            lut[((size_t)r * LN + b) * 2] = (uint32_t)j | ((uint32_t)row[j] << 16);
            // a candidate that is the row's escape slot gets frequency 0: the decoder's one range check then also routes
            // escapes away from its fast path
            lut[((size_t)r * LN + b) * 2 + 1] = j == len - 2 ? 0u : (uint32_t)(row[j + 1] - row[j]);
        }
    }
    // encoder entries: m = ceil(2^(63+s) / freq), s = ceil(log2 freq): floor(x * m / 2^(63+s)) == x / freq for every
    // x < 2^63 (the coder keeps x < freq << 47); freq == 1 uses m = 2^64 - 1 (q = x - 1) with the bias making up for it
    std::vector<uint32_t> enc((size_t)total * 4);
    for (int r = 0; r < nrows; ++r) {
        const int32_t* row = cdf + (size_t)r * stride;
        for (int j = 0; j < sizes[r] - 1; ++j) {
            const uint32_t start = (uint32_t)row[j], freq = (uint32_t)(row[j + 1] - row[j]);
            uint64_t m;
            uint32_t shift, bias;
            if (freq == 1) {
                m = ~0ull;
                shift = 0;
                bias = start + 65535u;
            } else {
                int sl = 0;
                while ((1u << sl) < freq) ++sl;
                const unsigned __int128 num = ((unsigned __int128)1 << (63 + sl)) + freq - 1;
                m = (uint64_t)(num / freq);
                shift = (uint32_t)(sl - 1);
                bias = start;
            }
            uint32_t* e = &enc[((size_t)row_off[r] + j) * 4];
            e[0] = (uint32_t)m;
            e[1] = (uint32_t)(m >> 32);
            e[2] = bias | (shift << 17);
            e[3] = freq;
        }
    }
    // decoder probe array: every row again as cdf - 1 (entry 0: 0) followed by 64 pad entries 0xFFFF (= 65536 - 1), so a
    // 64-wide "entry < cum" probe needs no bounds and a probe that ends on the pad has found the row's escape slot
    std::vector<uint16_t> cm((size_t)total + 64 * (size_t)nrows, (uint16_t)0xFFFFu);
    for (int r = 0; r < nrows; ++r) {
        const int32_t* row = cdf + (size_t)r * stride;
        for (int j = 0; j < sizes[r] - 1; ++j) cm[(size_t)row_off[r] + 64 * (size_t)r + j] = (uint16_t)(j ? row[j] - 1 : 0);
    }
    // first-level probe rows: slot i of a row's first 64 as {0xFFFF - cdf[i] << 16 | 0xFFFF - (cdf[i + 1] - 1)}.  One 16-bit
    // compare of the low halves against 0xFFFF - cum counts the symbols below cum and ONE lane read then yields start and
    // end of the symbol.  A low half of 0 means "not resolved here": the row's last (escape) slot, the pad behind it, and
    // slot 63 of a row wider than the 64 lanes (the decoder sends index 63 to the bucket table, which knows which it is).
    std::vector<uint32_t> pk((size_t)nrows * 64, 0u);
    for (int r = 0; r < nrows; ++r) {
        const int32_t* row = cdf + (size_t)r * stride;
        const int n = sizes[r] - 1;  // slots
        for (int j = 0; j < n && j < 64; ++j) {
            if (j == 63 && n > 64) {  // the rest of a wide row: the identity step (freq 65536, start 0)
                pk[(size_t)r * 64 + j] = 0xFFFF0000u;
                continue;
            }
            pk[(size_t)r * 64 + j] = ((0xFFFFu - (uint32_t)row[j]) << 16) | (0x10000u - (uint32_t)row[j + 1]);
        }
    }
    // coarse first level of the rows with 129 ... 4032 slots (the decoder's loop for batches with several symbols on such
    // rows): slot j = the block of `stride` symbols from j * stride on, {cdf[first] << 16 | 0x10000 - cdf[end]} (the same
    // 16-bit compare that resolves a narrow symbol yields the block; blocks behind the row's end never compare), followed
    // in the decoder by ONE 64-wide probe of the block in the cdf - 1 array above.  Narrower wide rows stay with the bucket
    // table: it resolves them in one hop.
    std::vector<int32_t> coarse(nrows, -1);
    std::vector<uint32_t> pkc;
    // (the decoder's tables without any coarse row, as rans_decode_lds_bytes counts them: a table set that fits the LDS
    // without coarse rows must keep fitting -- rows that would not fit stay with the bucket table)
    const size_t lds_base = (((size_t)nrows * LN + 1) & ~(size_t)1) * 8 + (((size_t)nrows + 1) & ~(size_t)1) * 8 +
                            ((((size_t)total + 64 * (size_t)nrows) * 2 + 15) & ~(size_t)15) + (size_t)nrows * 256;
    const size_t lds_room = lds_base < 157 * 1024 ? (157 * 1024 - lds_base) / 256 : 0;
    for (int r = 0; r < nrows; ++r) {
        const int32_t* row = cdf + (size_t)r * stride;
        const int n = sizes[r] - 1;
        if (n <= 128 || n > 4032 || pkc.size() / 64 >= 255 || pkc.size() / 64 >= lds_room) continue;
        coarse[r] = (int32_t)(pkc.size() / 64);
        pkc.resize(pkc.size() + 64, 0u);
        const int st = (n + 63) / 64;
        for (int j = 0; j < 64 && j * st < n; ++j) {
            const int s0 = j * st, e = std::min(s0 + st, n);
            pkc[(size_t)coarse[r] * 64 + j] = ((uint32_t)row[s0] << 16) | ((0x10000u - (uint32_t)row[e]) & 0xFFFFu);
        }
    }
    const size_t b_cm = (cm.size() * 2 + 15) & ~(size_t)15;
    const size_t b_pk = pk.size() * 4;
    const size_t b_pkc = pkc.size() * 4;
    const size_t b_enc = (size_t)total * 16;
    const size_t b_cdf = ((size_t)total * 2 + 15) & ~(size_t)15;
    const size_t b_lut = ((size_t)nrows * LN * 8 + 15) & ~(size_t)15;
    const size_t b_i32 = ((size_t)nrows * 4 + 15) & ~(size_t)15;
    const size_t bytes = b_cdf + b_lut + 3 * b_i32 + b_enc + b_cm + b_pk + b_pkc + b_i32;
    ts->blob = nullptr;  // a previous blob stays with its owner (TableSet::hold / rgbd_tables_destroy)
    ts->ready = false;
    HIP_TRY(hipMalloc(&ts->blob, bytes));
    std::vector<unsigned char> host(bytes, 0);
    unsigned char* p = host.data();
    memcpy(p, packed.data(), (size_t)total * 2);
    memcpy(p + b_cdf, lut.data(), (size_t)nrows * LN * 8);
    memcpy(p + b_cdf + b_lut, row_off.data(), (size_t)nrows * 4);
    memcpy(p + b_cdf + b_lut + b_i32, sizes, (size_t)nrows * 4);
    memcpy(p + b_cdf + b_lut + 2 * b_i32, offsets, (size_t)nrows * 4);
    memcpy(p + b_cdf + b_lut + 3 * b_i32, enc.data(), b_enc);
    memcpy(p + b_cdf + b_lut + 3 * b_i32 + b_enc, cm.data(), cm.size() * 2);
    memcpy(p + b_cdf + b_lut + 3 * b_i32 + b_enc + b_cm, pk.data(), b_pk);
    if (b_pkc) memcpy(p + b_cdf + b_lut + 3 * b_i32 + b_enc + b_cm + b_pk, pkc.data(), b_pkc);
    memcpy(p + b_cdf + b_lut + 3 * b_i32 + b_enc + b_cm + b_pk + b_pkc, coarse.data(), (size_t)nrows * 4);
    HIP_TRY(hipMemcpy(ts->blob, host.data(), bytes, hipMemcpyHostToDevice));
    unsigned char* dp = (unsigned char*)ts->blob;
    ts->d.cdf = (const uint16_t*)dp;
    ts->d.lut = (const uint32_t*)(dp + b_cdf);
    ts->d.lut_bits = bits;
    ts->d.row_off = (const int32_t*)(dp + b_cdf + b_lut);
    ts->d.sizes = (const int32_t*)(dp + b_cdf + b_lut + b_i32);
    ts->d.offsets = (const int32_t*)(dp + b_cdf + b_lut + 2 * b_i32);
    ts->d.enc = (const uint32_t*)(dp + b_cdf + b_lut + 3 * b_i32);
    ts->d.cm = (const uint16_t*)(dp + b_cdf + b_lut + 3 * b_i32 + b_enc);
    ts->d.pk = (const uint32_t*)(dp + b_cdf + b_lut + 3 * b_i32 + b_enc + b_cm);
    ts->d.pkc = (const uint32_t*)(dp + b_cdf + b_lut + 3 * b_i32 + b_enc + b_cm + b_pk);
    ts->d.coarse = (const int32_t*)(dp + b_cdf + b_lut + 3 * b_i32 + b_enc + b_cm + b_pk + b_pkc);
    ts->d.ncoarse = (int)(pkc.size() / 64);
    ts->d.nrows = nrows;
    ts->d.total = total;
    ts->ready = true;
    ts->stride_src = stride;
    return RGBD_OK;
}

// ---- checkerboard quantise / index ----------------------------------------------------------------------------------------
namespace {

struct ScaleTable64 {
    float v[64];
};

// One checkerboard half of an NCHW slice in ONE pass, in place on the caller's tensors (no staging copies): thread i owns the
// packed position (n, c, row, k) -- the order of the reference's .reshape(-1) of the squeezed tensor (ckbd.py:83-105) -- i.e.
// the column pair (2k, 2k + 1) of that row.  VEC: the pair is read / written as one 8-byte access (every row starts on an even
// element because w is even; needs 8-byte aligned tensors), so a wavefront's loads cover 512 contiguous bytes per operand.
// MODE 0: symbol = rint(y - mean), index = build_indexes(scale), y_hat = symbol + mean (entropy_models.py:118-146,561-568);
// MODE 2: y_hat = symbol + mean.  The anchor pass also writes the zero of the pair's other column (ckbd.py:66-72).
// Same operations, in the same order, as ckbd_part_kernel (entropy.hip) performs on the codec's own layout.
template <int MODE, bool VEC>
__global__ void ckbd_nchw_kernel(const float* __restrict__ y, const float* __restrict__ means, const float* __restrict__ scales,
                                 float* __restrict__ yhat, ScaleTable64 table, int rows, int h, int w2, int anchor,
                                 int32_t* __restrict__ sym, int32_t* __restrict__ idx)
{
    __shared__ float tbl[64];
    if (threadIdx.x < 64) tbl[threadIdx.x] = table.v[threadIdx.x];
    __syncthreads();
    const size_t total = (size_t)rows * w2;  // rows = n * c * h
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / w2;
        const int k = (int)(i - r * w2);
        const int row = (int)(r % h);
        const int par = ckbd_col(row, 0, anchor);  // 0 / 1: which column of the pair this half codes
        const size_t p0 = (r * w2 + k) * 2;        // element index of column 2k
        float mean, scale = 0.f, yv = 0.f;
        if (VEC) {
            const float2 m2 = *reinterpret_cast<const float2*>(means + p0);
            mean = par ? m2.y : m2.x;
            if (MODE == 0) {
                const float2 y2 = *reinterpret_cast<const float2*>(y + p0);
                const float2 s2 = *reinterpret_cast<const float2*>(scales + p0);
                yv = par ? y2.y : y2.x;
                scale = par ? s2.y : s2.x;
            }
        } else {
            mean = means[p0 + par];
            if (MODE == 0) {
                yv = y[p0 + par];
                scale = scales[p0 + par];
            }
        }
        int s;
        if (MODE == 0) {
            s = (int)rintf(yv - mean);  // round half to even, like torch.round
            sym[i] = s;
            idx[i] = scale_to_index(tbl, scale);
        } else {
            s = sym[i];
        }
        const float out = (float)s + mean;
        if (anchor) {
            if (VEC) {
                *reinterpret_cast<float2*>(yhat + p0) = par ? make_float2(0.f, out) : make_float2(out, 0.f);
            } else {
                yhat[p0 + par] = out;
                yhat[p0 + (par ^ 1)] = 0.f;
            }
        } else {
            yhat[p0 + par] = out;
        }
    }
}

int ckbd_op(int mode, const float* y_dev, const float* means_dev, const float* scales_dev, int32_t n, int32_t c, int32_t h, int32_t w,
            int32_t anchor, const float* scale_table, int32_t* symbols_dev, int32_t* indexes_dev, float* yhat_dev, void* stream)
{
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || (w & 1) || !means_dev || !yhat_dev || !symbols_dev) return RGBD_EINVAL;
    if (mode == 0 && (!y_dev || !scales_dev || !scale_table || !indexes_dev)) return RGBD_EINVAL;
    if ((int64_t)n * c * h * w >= ((int64_t)1 << 31)) return RGBD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ScaleTable64 tb{};
    if (mode == 0) memcpy(tb.v, scale_table, sizeof(tb.v));
    const int rows = n * c * h, w2 = w / 2;
    const size_t total = (size_t)rows * w2;
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 8192);  // (grid-stride: ~32 workgroups per CU at most)
    auto al8 = [](const void* p) { return ((uintptr_t)p & 7u) == 0; };
    const bool vec = al8(means_dev) && al8(yhat_dev) && (mode != 0 || (al8(y_dev) && al8(scales_dev)));
    const int an = anchor ? 1 : 0;
    if (mode == 0) {
        if (vec) hipLaunchKernelGGL((ckbd_nchw_kernel<0, true>), dim3(grid), dim3(256), 0, s, y_dev, means_dev, scales_dev, yhat_dev, tb, rows, h, w2, an, symbols_dev, indexes_dev);
        else hipLaunchKernelGGL((ckbd_nchw_kernel<0, false>), dim3(grid), dim3(256), 0, s, y_dev, means_dev, scales_dev, yhat_dev, tb, rows, h, w2, an, symbols_dev, indexes_dev);
    } else {
        if (vec) hipLaunchKernelGGL((ckbd_nchw_kernel<2, true>), dim3(grid), dim3(256), 0, s, y_dev, means_dev, scales_dev, yhat_dev, tb, rows, h, w2, an, symbols_dev, indexes_dev);
        else hipLaunchKernelGGL((ckbd_nchw_kernel<2, false>), dim3(grid), dim3(256), 0, s, y_dev, means_dev, scales_dev, yhat_dev, tb, rows, h, w2, an, symbols_dev, indexes_dev);
    }
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

}  // namespace

extern "C" {


// ops.cpp:24-81 restated (host, one-off table construction)
int rgbd_pmf_to_quantized_cdf(const float* pmf, int32_t n, int32_t precision, uint32_t* cdf_out)
{
    if (!pmf || !cdf_out || n <= 0 || precision < 1 || precision > 16) return RGBD_EINVAL;
    std::vector<uint32_t> c((size_t)n + 1);
    c[0] = 0;
    const float scale = (float)(1 << precision);
    for (int i = 0; i < n; ++i) c[(size_t)i + 1] = (uint32_t)std::round(pmf[i] * scale);
    uint32_t total = 0;
    for (uint32_t v : c) total += v;
    if (!total) return RGBD_EINVAL;
    for (uint32_t& v : c) v = (uint32_t)((((uint64_t)1 << precision) * v) / total);
    for (size_t i = 1; i < c.size(); ++i) c[i] += c[i - 1];
    c.back() = 1u << precision;
    const int m = n + 1;
    for (int i = 0; i < m - 1; ++i) {
        if (c[i] != c[i + 1]) continue;
        uint32_t best = ~0u;
        int donor = -1;
        for (int j = 0; j < m - 1; ++j) {
            const uint32_t f = c[j + 1] - c[j];
            if (f > 1 && f < best) {
                best = f;
                donor = j;
            }
        }
        if (donor < 0) return RGBD_EINVAL;
        if (donor < i)
            for (int j = donor + 1; j <= i; ++j) c[j]--;
        else
            for (int j = i + 1; j <= donor; ++j) c[j]++;
    }
    memcpy(cdf_out, c.data(), sizeof(uint32_t) * c.size());
    return RGBD_OK;
}

int rgbd_tables_create(const int32_t* cdf, int32_t cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets,
                       int32_t n_cdf, rgbd_tables** out)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!out) return RGBD_EINVAL;
    std::unique_ptr<rgbd_tables> t(new rgbd_tables());
    const int r = build_tables(cdf, cdf_stride, cdf_sizes, offsets, n_cdf, &t->ts);
    if (r) {
        if (t->ts.blob) (void)hipFree(t->ts.blob);
        return r;
    }
    *out = t.release();
    return RGBD_OK;
}

void rgbd_tables_destroy(rgbd_tables* t)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!t) return;
    if (t->ts.blob) (void)hipFree(t->ts.blob);
    delete t;
}

static int64_t enc_cap_words(int64_t n) { return rgbd_enc_cap_words(n); }

int64_t rgbd_rans_max_bytes(int64_t n) { return 4 * enc_cap_words(n); }

int rgbd_rans_encode(const rgbd_tables* t, const int32_t* symbols, const int32_t* indexes, int64_t n, uint8_t* out,
                     int64_t cap, int64_t* out_len)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!t || !t->ts.ready || n < 0 || !out || !out_len || (n && (!symbols || !indexes))) return RGBD_EINVAL;
    for (int64_t i = 0; i < n; ++i)
        if (indexes[i] < 0 || indexes[i] >= t->ts.d.nrows) return RGBD_EINVAL;
    const int64_t capw = enc_cap_words(n);
    int32_t *dsym = nullptr, *didx = nullptr;
    uint32_t* dout = nullptr;
    int64_t* dmeta = nullptr;
    int* derr = nullptr;
    int rc = RGBD_OK;
    auto cleanup = [&]() {
        (void)hipFree(dsym);
        (void)hipFree(didx);
        (void)hipFree(dout);
        (void)hipFree(dmeta);
        (void)hipFree(derr);
    };
    HIP_TRY(hipMalloc((void**)&dsym, sizeof(int32_t) * (size_t)(n + 1)));
    HIP_TRY(hipMalloc((void**)&didx, sizeof(int32_t) * (size_t)(n + 1)));
    HIP_TRY(hipMalloc((void**)&dout, sizeof(uint32_t) * (size_t)capw));
    HIP_TRY(hipMalloc((void**)&dmeta, sizeof(int64_t) * 4));
    HIP_TRY(hipMalloc((void**)&derr, sizeof(int)));
    const int64_t hmeta[4] = {0, n, 0, 0};
    hipError_t e = hipSuccess;
    if (n) {
        e = hipMemcpy(dsym, symbols, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(didx, indexes, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemcpy(dmeta, hmeta, sizeof(hmeta), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(derr, 0, sizeof(int));
    if (e != hipSuccess) {
        cleanup();
        return RGBD_EHIP;
    }
    rc = launch_rans_encode(dsym, didx, dmeta, dmeta + 1, 1, 1, t->ts.d, t->ts.d, dout, capw, dmeta + 2, derr, nullptr);
    int64_t nw = 0;
    int herr = 0;
    if (!rc) {
        e = hipMemcpy(&nw, dmeta + 2, sizeof(int64_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&herr, derr, sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = RGBD_EHIP;
        else if (herr) rc = RGBD_ENOSPC;
        else if (nw * 4 > cap) rc = RGBD_ENOSPC;
        else {
            e = hipMemcpy(out, dout + (capw - nw), (size_t)nw * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = RGBD_EHIP;
            *out_len = nw * 4;
        }
    }
    cleanup();
    return rc;
}

struct rgbd_rans_decoder {
    uint32_t* words = nullptr;
    int64_t nwords = 0;
    int64_t* meta = nullptr;   // [off, len, base]
    uint64_t* state = nullptr;  // [x, pos]
    bool fresh = false;
};

int rgbd_rans_decoder_create(rgbd_rans_decoder** out)
{
    if (!out) return RGBD_EINVAL;
    std::unique_ptr<rgbd_rans_decoder> d(new rgbd_rans_decoder());
    HIP_TRY(hipMalloc((void**)&d->meta, sizeof(int64_t) * 4));
    HIP_TRY(hipMalloc((void**)&d->state, sizeof(uint64_t) * 2));
    *out = d.release();
    return RGBD_OK;
}

int rgbd_rans_decoder_set_stream(rgbd_rans_decoder* d, const uint8_t* stream, int64_t nbytes)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!d || !stream || nbytes < 8 || (nbytes & 3)) return RGBD_EINVAL;
    if (d->words) (void)hipFree(d->words);
    d->words = nullptr;
    HIP_TRY(hipMalloc((void**)&d->words, (size_t)nbytes));
    HIP_TRY(hipMemcpy(d->words, stream, (size_t)nbytes, hipMemcpyHostToDevice));
    d->nwords = nbytes / 4;
    const int64_t hm[4] = {0, d->nwords, 0, 0};
    HIP_TRY(hipMemcpy(d->meta, hm, sizeof(hm), hipMemcpyHostToDevice));
    d->fresh = true;
    return RGBD_OK;
}

int rgbd_rans_decoder_decode(rgbd_rans_decoder* d, const rgbd_tables* t, const int32_t* indexes, int64_t n,
                             int32_t* symbols_out)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!d || !d->words || !t || !t->ts.ready || n < 0 || (n && (!indexes || !symbols_out))) return RGBD_EINVAL;
    if (!n) return RGBD_OK;
    for (int64_t i = 0; i < n; ++i)
        if (indexes[i] < 0 || indexes[i] >= t->ts.d.nrows) return RGBD_EINVAL;
    int32_t *didx = nullptr, *dsym = nullptr;
    HIP_TRY(hipMalloc((void**)&didx, sizeof(int32_t) * (size_t)n));
    HIP_TRY(hipMalloc((void**)&dsym, sizeof(int32_t) * (size_t)n));
    int rc = RGBD_OK;
    if (hipMemcpy(didx, indexes, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) rc = RGBD_EHIP;
    if (!rc)
        rc = launch_rans_decode(d->words, d->meta, d->meta + 1, 1, d->state, d->fresh ? 1 : 0, didx, dsym, d->meta + 2, 0, n,
                                t->ts.d, nullptr);
    if (!rc && hipMemcpy(symbols_out, dsym, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
        rc = RGBD_EHIP;
    if (!rc) d->fresh = false;
    (void)hipFree(didx);
    (void)hipFree(dsym);
    return rc;
}

void rgbd_rans_decoder_destroy(rgbd_rans_decoder* d)
{
    std::unique_lock<std::shared_mutex> cap_lk(g_capture_mu);  // frees / synchronous copies: not while a stream captures
    if (!d) return;
    (void)hipFree(d->words);
    (void)hipFree(d->meta);
    (void)hipFree(d->state);
    delete d;
}

// ---- device-batched coder -------------------------------------------------------------------------------------------------
// The codec's own path (engine.hip run_compress / run_decompress) with the pointers handed in by the caller: nothing is
// copied, nothing is allocated, the launch is asynchronous on `stream`.
int rgbd_rans_encode_batch_dev(const rgbd_tables* t, const int32_t* symbols_dev, const int32_t* indexes_dev,
                               const int64_t* sym_base_dev, const int64_t* counts_dev, int32_t nstreams, uint32_t* out_dev,
                               int64_t cap_words, int64_t* out_words_dev, int32_t* err_dev, void* stream)
{
    if (!t || !t->ts.ready || nstreams < 0 || !sym_base_dev || !counts_dev || !out_dev || !out_words_dev || !err_dev)
        return RGBD_EINVAL;
    if (nstreams && (!symbols_dev || !indexes_dev)) return RGBD_EINVAL;
    if (cap_words < 64 || cap_words % 64) return RGBD_EINVAL;
    return launch_rans_encode(symbols_dev, indexes_dev, sym_base_dev, counts_dev, nstreams, nstreams, t->ts.d, t->ts.d, out_dev,
                              cap_words, out_words_dev, err_dev, (hipStream_t)stream);
}

int rgbd_rans_decode_batch_dev(const rgbd_tables* t, const uint32_t* streams_dev, const int64_t* stream_off_words_dev,
                               const int64_t* stream_len_words_dev, int32_t nstreams, uint64_t* state_dev, int32_t init,
                               const int32_t* indexes_dev, int32_t* symbols_dev, const int64_t* sym_base_dev, int64_t part_off,
                               int64_t count, void* stream)
{
    if (!t || !t->ts.ready || nstreams < 0 || count < 0 || part_off < 0) return RGBD_EINVAL;
    if (!nstreams || !count) return RGBD_OK;
    if (!streams_dev || !stream_off_words_dev || !stream_len_words_dev || !state_dev || !indexes_dev || !symbols_dev || !sym_base_dev)
        return RGBD_EINVAL;
    return launch_rans_decode(streams_dev, stream_off_words_dev, stream_len_words_dev, nstreams, state_dev, init ? 1 : 0, indexes_dev,
                              symbols_dev, sym_base_dev, part_off, count, t->ts.d, (hipStream_t)stream);
}

// ---- checkerboard quantise / index (ckbd_op above) ----

int rgbd_ckbd_quant_index(const float* y_dev, const float* means_dev, const float* scales_dev, int32_t n, int32_t c, int32_t h,
                          int32_t w, int32_t anchor, const float* scale_table, int32_t* symbols_dev, int32_t* indexes_dev,
                          float* yhat_dev, void* stream)
{
    return ckbd_op(0, y_dev, means_dev, scales_dev, n, c, h, w, anchor, scale_table, symbols_dev, indexes_dev, yhat_dev, stream);
}

int rgbd_ckbd_dequant(const int32_t* symbols_dev, const float* means_dev, int32_t n, int32_t c, int32_t h, int32_t w, int32_t anchor,
                      float* yhat_dev, void* stream)
{
    return ckbd_op(2, nullptr, means_dev, nullptr, n, c, h, w, anchor, nullptr, const_cast<int32_t*>(symbols_dev), nullptr, yhat_dev,
                   stream);
}

}  // extern "C"
