// Shared declarations for the gfx950 ELIC_united engine (internal; the public C ABI is include/rgbd_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define RGBD_OK 0
#define RGBD_EINVAL (-22)
#define RGBD_ENOMEM (-12)
#define RGBD_EHIP (-5)
#define RGBD_ENOSPC (-28)
#define RGBD_ESTATE (-1)

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            fprintf(stderr, "[rgbd_amd] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr,          \
                    hipGetErrorString(_e));                                                    \
            return RGBD_EHIP;                                                                  \
        }                                                                                      \
    } while (0)

// Activation tensor in device memory: NHWC fp32, channel stride cs (multiple of 16, pad channels hold zeros).
struct Act {
    float* p = nullptr;
    int n = 0, h = 0, w = 0, c = 0, cs = 0;
    size_t elems() const { return (size_t)n * h * w * cs; }
};

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Physical position of logical channel c inside its group of 16 (engines with rgbd_elic::chperm; a 4x4 transpose, its own
// inverse).  The MFMA kernels reduce a 16-channel chunk k-step by k-step, lane group by lane group: physical channels
// e, 4+e, 8+e, 12+e for e = 0..3.  With logical channel 4e + q stored at physical 4q + e that order IS channel 0, 1, ... 15 --
// the order the reference's CPU kernels accumulate in -- without touching a kernel: weights are packed with
// both channel axes permuted, and only code that tells channels apart (bias / gate vectors, quantisers, NCHW conversion) maps.
__host__ __device__ static inline int rgbd_cperm(int c, int on = 1) { return on ? ((c & ~15) | ((c & 3) << 2) | ((c >> 2) & 3)) : c; }

// ---- convolution launcher (conv_mfma.hip) -----------------------------------------------------
struct TapTable {
    int8_t dy[4][25];
    int8_t dx[4][25];
    int8_t wt[4][25];
    int8_t n[4];
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_SIGMOID = 3, ACT_GELU = 4 };

// Second operand set of a grouped launch (ConvArgs::groups == 2): the same layer shape on independent data with its own
// weights -- the RGB and the depth branch of the transforms run as ONE launch with twice the workgroups.  Every stride,
// extent and flag is shared; a pointer is null here exactly when its ConvArgs twin is.
struct ConvPtrs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    const float* res1;
    const float* mul;
    const float* res2;
    float* partial;
    float* y2;
    const float* w2;
    const float* bias2;
    const float* w3;
    const float* bias3;
    float* y3;
};

struct ConvArgs {
    const float* x;  // input NHWC (already offset to the first input channel)
    int N, H, W, xcs;
    int cin_pad;  // channels reduced over (multiple of 16)
    const float* w;  // packed [cout_pad][ntaps_total][cin_pad]
    int ntaps_total;
    const float* bias;  // [cout_pad]
    float* y;  // output NHWC (already offset to the first output channel)
    int OH, OW, ycs;
    int cout_pad;  // channels computed (multiple of 16)
    int cout_store;  // channels written (multiple of 4, <= cout_pad; 0 = cout_pad): narrower when the output is a channel
                     // slice whose width is not a multiple of 16 (the tail would land in the neighbouring slice)
    int GH, GW;  // tile-grid extent (output positions per phase)
    int IS, OS;  // input step / output step per grid position
    int nphase;  // 1, or 4 for stride-2 transposed conv (phase = py*2+px)
    int min_dy, min_dx, span_y, span_x;
    int act;
    const float* res1;  // added before the activation
    int r1cs;
    const float* mul;  // multiplied after the activation
    int mcs;
    const float* res2;  // added last
    int r2cs;
    int splitk;      // >1: reduce the input channels in `splitk` fixed ranges (partials + ordered sum); 0/1 = off
    float* partial;  // [splitk][N*OH*OW][cout_pad] scratch when splitk > 1
    int loaded;      // tile-table flavour: 0 = winners of isolated launches (latency), 1 = winners with the chip shared (throughput)
    int ckbd;        // checkerboard output: 0 = every position, 1 = anchor positions only ((row + col) odd, ckbd.py:37-48),
                     // 2 = non-anchor positions only; the other half of y is left untouched (stride-1, single-phase convs)
    float* y2;       // optional second destination of the same output (another concat buffer), channel stride y2cs
    int y2cs;
    int subpix;      // 1: sub-pixel form of a stride-2 transposed conv with <= 4 couts: one stride-1 conv over the input grid
                     // whose 16 channels are (output phase py*2+px) * 4 + channel, each float4 stored to its own output pixel
    // Fused trailing 1x1 (launch_conv_fused; ResidualBottleneck branch.2 -> branch.4, ResidualUnit conv.2 -> conv.4):
    // t = act_mid(conv(x) + bias) stays in the accumulator registers and is the B operand of a second GEMM
    // y = act(w2 * t + bias2 + res1).  cout_pad / w / bias describe the first layer, y / ycs / cout_store / res1 / act the
    // second; the fma chain of every y element is the one the stand-alone 1x1 launch runs (chunk -> channel).
    const float* w2;     // packed [cout2_pad][1][cout_pad]; nullptr = no fused layer
    const float* bias2;  // [cout2_pad]
    int cout2_pad;
    int act_mid;
    // ... and, optionally, the leading 1x1 of the block that follows: u = relu(w3 * y + bias3) -> y3 (cout3_pad = cout_pad)
    const float* w3;     // packed [cout3_pad][1][cout2_pad]; nullptr = none
    const float* bias3;
    float* y3;
    int y3cs;
    int cout3_pad;
    int groups;   // 0 / 1: one operand set; 2: workgroups [tiles*N, 2*tiles*N) of the work list run the same layer on g1
    ConvPtrs g1;
    TapTable taps;
    // ---- reference arithmetic: the accumulation structure of the CPU kernels the reference runs on ----------
    int blocked;         // 1: blocked accumulation (conv_mfma_blk.hip): a fresh fma chain per block, block sums added in order
    int bias_mode;       // where the bias enters: 0 the epilogue (sum, then + bias), 1 the running total starts from it
                         // (total = S_0 + bias: oneDNN's direct kernels), 2 the first chain starts from it (its 1x1 kernels)
    int tail_bias_init;  // fused trailing / leading 1x1 layers (w2, w3): their chains start from the bias (as bias_mode 2)
    int exact_math;      // epilogue sigmoid as the reference's vectorised CPU kernel computes it (Sleef expf_u10 + IEEE divide)
    uint32_t blk_end[8];      // blocked: bit c set = a block ends with 16-channel chunk c (the last chunk's bit is always set)
    uint16_t split_c16[18];   // splitk > 1 with explicit ranges: split s reduces chunks [split_c16[s], split_c16[s + 1]); all 0 = even ranges
};

// split factor of a layer: a function of the layer and of the per-image output grid only -- never of the batch size -- so
// the summation order of every output is the same in the encoder, the decoder and for any batching of the same images
extern char g_conv_force[64];
extern int g_fuse_force, g_fuse_lead_off;  // rgbd_debug_force_fuse (conv_mfma.hip)
// measured exceptions to the rule below (tools/tune_splitk.py): {out_px_per_image, cin_pad, cout_pad, taps per phase, nphase, S}
struct SplitKEntry {
    int out_px, cin_pad, cout_pad, ntaps, nphase, s;
};
static const SplitKEntry kSplitKTable[] = {
#include "splitk_table.h"
    {0, 0, 0, 0, 0, 0}};
// 0: the layer shape is not in the table
static inline int conv_splitk_table(int cin_pad, int cout_pad, int taps_per_phase, long out_px_per_image, int nphase)
{
    for (const SplitKEntry* e = kSplitKTable; e->out_px; ++e)
        if (e->out_px == out_px_per_image && e->cin_pad == cin_pad && e->cout_pad == cout_pad && e->ntaps == taps_per_phase &&
            e->nphase == nphase)
            return e->s;
    return 0;
}
static inline int conv_splitk_for(int cin_pad, int cout_pad, int taps_per_phase, long out_px_per_image, int nphase)
{
    if (const int t = conv_splitk_table(cin_pad, cout_pad, taps_per_phase, out_px_per_image, nphase)) return t;
    const long K = (long)cin_pad * taps_per_phase;
    int s;
    if (nphase == 1 && out_px_per_image <= 512) {
        // 16x16-class latent grids (tools/tile_sweep.py): few output tiles, so the reduction is cut until every CU has
        // work; 1x1 layers (pure weight streaming) profit from the deepest cut
        if (taps_per_phase == 1) s = K >= 1500 ? 8 : (K >= 700 ? 4 : (K >= 350 ? 2 : 1));
        else s = K >= 5000 ? 8 : (K >= 1000 ? 4 : (K >= 500 ? 2 : 1));
    } else {
        s = (int)((K + 800) / 1600);
        if (s > 8) s = 8;
    }
    if (s > cin_pad / 16) s = cin_pad / 16;
    if (s < 1) s = 1;
    return s;
}

int launch_conv(const ConvArgs& a, hipStream_t s);
// conv + fused trailing 1x1 (ConvArgs::w2).  conv_fused_plan returns the pixel-tile class the launch would use
// (1 / 2 / 4 = 64 / 128 / 256 pixels per workgroup) or 0 when this pair of layers on this grid is not worth fusing.
int conv_fused_plan(int cout_pad, int cout2_pad, int ntaps, int N, int GH, int GW, int loaded);
int launch_conv_fused(const ConvArgs& a, hipStream_t s);
int conv_log_enable(int on);
int conv_tile_override(const char* csv);  // in-situ tile overrides by shape key (tools/tune_insitu.py); nullptr / "" clears
long conv_log_read(char* buf, long cap);  // CSV text; returns the size needed

// y = mul * sigmoid(t) + res2 with the reference's per-element choice of sigmoid form (vector / scalar tail of torch's CPU loop)
int launch_sigmoid_gate_ref(const float* t, int tcs, const float* mul, int mcs, const float* res2, int r2cs, float* y, int ycs,
                            int N, int HW, int C, int per_image, int threads, hipStream_t s);

// ---- pointwise kernels (pointwise.hip) --------------------------------------------------------
int launch_nchw_to_nhwc16(const float* src, int N, int C, int H, int W, float* dst, int cs, hipStream_t s, int perm = 0);
int launch_nhwc_to_nchw_clamp(const float* src, int N, int C, int H, int W, int cs, float* dst, int clamp01,
                              hipStream_t s, int perm = 0);
// x1 / y1 (both or neither): a second tensor pair of the same shape handled by the same launch (the other modality's ESA branch)
int launch_maxpool7s3(const float* x, int N, int H, int W, int cs, float* y, int OH, int OW, hipStream_t s,
                      const float* x1 = nullptr, float* y1 = nullptr);
// ref_channels > 0: the reference's CPU arithmetic (see bilinear_kernel) for a tensor of that many channels, stored permuted
int launch_bilinear(const float* x, int N, int h, int w, int cs, float* y, int H, int W, hipStream_t s, const float* x1 = nullptr,
                    float* y1 = nullptr, int ref_channels = 0);
int launch_channel_mean(const float* x, int N, int HW, int cs, int C, float* mean, hipStream_t s);
int launch_channel_mean_strided(const float* x, int N, int HW, int cs, int C, float* mean, int mstride, hipStream_t s);
int launch_channel_scale_to_strided(const float* x, int N, int HW, int xcs, int C, const float* scale, int sstride, int mode,
                                    float* y, int ycs, hipStream_t s);
// w1t is fc.2.weight transposed to [hidden][C]; hid is a [N][hidden] scratch buffer
// mstride: distance between the images' mean vectors (0 = C: packed)
// perm: the mean / scale vectors are indexed by channel POSITION (rgbd_cperm), the FC weights by channel
int launch_se_fc(const float* mean, int N, int C, int hidden, const float* w0, const float* w1t, float* hid,
                 float* scale, hipStream_t s, int mstride = 0, int perm = 0);
// the same block in the reference's CPU arithmetic (pointwise.hip: channel_mean_ref_kernel, se_linear_ref_kernel).  w1 is
// fc.2.weight as stored ([C][hidden]); cls0 / cls1: per-row dot-product class of the two layers (device, nullptr = all main)
int launch_channel_mean_ref(const float* x, int N, int HW, int cs, int C, float* mean, int mstride, hipStream_t s);
int launch_se_fc_ref(const float* mean, int N, int C, int hidden, const float* w0, const float* w1, const int* cls0,
                     const int* cls1, float* hid, float* scale, hipStream_t s, int mstride = 0, int form = -1);
// mode 0: y = x*s ; mode 1: y = x + x*s   (s per (n, c))
int launch_channel_scale_to(const float* x, int N, int HW, int xcs, int C, const float* scale, int mode, float* y, int ycs,
                            hipStream_t s);
int launch_copy_channels(const float* src, int scs, float* dst, int dcs, int npix, int C, hipStream_t s);
// small-tensor convolution in the reference's im2col + sgemm arithmetic (pointwise.hip: small_conv_ref_kernel)
struct SmallConvArgs {
    const float* x;
    const float* w;     // the MFMA kernels' packed weights [cout_pad][ntaps][cin_pad], read in place
    const float* bias;
    const float* res1;  // added before the activation (optional)
    const float* mul;   // multiplied after the activation (optional)
    const float* res2;  // added last (optional)
    float* y;
    float* y2;          // optional second destination
    int N, H, W, xcs, C, cin_pad, O, OH, OW, ycs, r1cs, mcs, r2cs, y2cs, K, stride, pad, act, nb;
    int ckbd;           // 1 / 2: only the anchor / non-anchor positions are computed (ConvArgs::ckbd)
    int kb[17];         // K-block boundaries in k = c * K * K + ky * K + kx
};
int launch_small_conv_ref(const SmallConvArgs& a, hipStream_t s);
// stride-2 transposed conv in the reference's arithmetic: the (tap, channel)-ordered GEMM of one (phase, column class)
int launch_gather_taps(const float* x, int B, int h, int w, int cs, int j0, int jw, int ntap, const int* dy, const int* dx,
                       float* col, hipStream_t s);
int launch_gather_wslabs(const float* wp, int cout_pad, int ntaps_total, int cin_pad, int ntap, const int* slab, float* wout,
                         hipStream_t s);
int launch_scatter_phase(const float* src, int B, int h, int jw, int scs, int j0, int py, int px, float* dst, int OW, int dcs,
                         int C, hipStream_t s);
int launch_fill_zero(float* p, size_t n, hipStream_t s);
// packed input of the first analysis conv: y[n][oy][ox][KP], see im2col5s2_kernel
int launch_im2col5s2(const float* x, int N, int H, int W, int cs, int C, float* y, int OH, int OW, int KP, hipStream_t s);
// swin.hip (STF_united)
// x1 / w1 / b1 / y1 (all or none): a second tensor of the same shape with its own weights in the same launch (the other modality)
int launch_layernorm(const float* x, size_t ntok, int C, int xcs, const float* w, const float* b, float* y, int ycs,
                     hipStream_t s, const float* x1 = nullptr, const float* w1 = nullptr, const float* b1 = nullptr,
                     float* y1 = nullptr);
int launch_window_attention(const float* qkv, int B, int H, int W, int C, int qcs, int heads, int shift, const float* rpb,
                            float* out, int ocs, hipStream_t s, const float* qkv1 = nullptr, const float* rpb1 = nullptr,
                            float* out1 = nullptr);
int launch_patch_merge_gather(const float* x, int B, int H, int W, int C, int xcs, float* y, int ycs, hipStream_t s);
int launch_pixel_shuffle2(const float* x, int B, int H, int W, int Co, int xcs, float* y, int ycs, hipStream_t s);

// ---- entropy stage (entropy.hip) --------------------------------------------------------------
struct DevTables {  // packed CDF rows for the device coder
    const uint16_t* cdf;  // concatenated rows, entries [0, size-1) (the final 65536 is implicit)
    const int32_t* row_off;  // [nrows] start of each row in cdf
    const int32_t* sizes;  // [nrows] reference cdf_length (= pmf_length + 2)
    const int32_t* offsets;  // [nrows]
    const uint32_t* lut;  // [nrows][2^lut_bits + 1][2]: {j | row[j] << 16, freq_j (0 if j is the escape slot)}, j = largest index with row[j] <= bucket start
    int lut_bits;
    int nrows;
    int total;  // total packed entries
    // encoder side, one entry per packed cdf entry (= per symbol of a row): exact division by the symbol frequency as a
    // multiply-high (Alverson; ryg_rans rans64.h:167-278 does the same for its Rans64EncSymbol):
    //   {m_lo, m_hi, bias | shift << 17, freq},  q = mulhi64(x, m) >> shift,  x' = x + bias + q * (65536 - freq)
    const uint32_t* enc;
    // decoder side: the rows as cdf - 1, each padded with 64 entries 0xFFFF; row r starts at row_off[r] + 64 * r
    const uint16_t* cm;
    // decoder, first level: [nrows][64] slots {0xFFFF - cdf[i] << 16 | 0x10000 - cdf[i + 1]} (low half 0: not resolved here)
    const uint32_t* pk;
    // decoder, coarse first level of the rows with 129 ... 4032 slots (at most 255 of them): [ncoarse][64] slots, slot j = the
    // block of stride = ceil(slots / 64) symbols from j * stride on, {cdf[first] << 16 | 0x10000 - cdf[end]}; coarse[r] =
    // the row's index in pkc or -1
    const uint32_t* pkc;
    const int32_t* coarse;
    int ncoarse;
};

struct PartGeom {
    int B, h, w;  // latent geometry
    int C;  // channels of this slice
    int anchor;  // 1 = anchor positions
    int per_image;  // 1: one stream per image; 0: one stream for the batch (reference B>1 format)
    int perm;       // 1: the tensors store channels at their permuted positions (rgbd_cperm); symbols stay in logical order
};

// checkerboard helpers: column of packed index k on a given row (utils/ckbd.py:51-64), and build_indexes
// (entropy_models.py:561-568) as a binary search of the 64-entry scale table
__device__ __forceinline__ int ckbd_col(int row, int k, int anchor)
{
    return 2 * k + (anchor ? (1 - (row & 1)) : (row & 1));
}

__device__ __forceinline__ int scale_to_index(const float* tbl, float s)
{
    // #{i < 63 : table[i] < max(s, 0.11)}  ==  63 - #{i < 63 : max(s, 0.11) <= table[i]}
    s = fmaxf(s, 0.11f);
    int lo = 0, hi = 63;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (tbl[mid] < s) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}


int launch_ckbd_encode_part(const float* y, int ycs, const float* params, int pcs, float* yhat, int yhcs,
                            const float* table, PartGeom g, int32_t* sym, int32_t* idx, const int64_t* stream_base,
                            int64_t part_off_per_image, hipStream_t s, float* dbg_x = nullptr, float* dbg_s = nullptr);
int launch_ckbd_index_part(const float* params, int pcs, const float* table, PartGeom g, int32_t* idx,
                           const int64_t* stream_base, int64_t part_off_per_image, hipStream_t s);
int launch_ckbd_decode_part(const float* params, int pcs, float* yhat, int yhcs, PartGeom g, const int32_t* sym,
                            const int64_t* stream_base, int64_t part_off_per_image, hipStream_t s);
int launch_ckbd_estimate_part(const float* y, int ycs, const float* params, int pcs, float* yhat, int yhcs, float* lik,
                              int lcs, PartGeom g, hipStream_t s);
// perm: z / zhat / lik store their channels permuted (rgbd_cperm); medians, prm and the symbol order are logical
int launch_eb_forward(const float* z, int zcs, int B, int h, int w, int C, const float* med, const float* prm, float* zhat,
                      float* lik, hipStream_t s, int perm = 0);
int launch_z_quant(const float* z, int zcs, int B, int h, int w, int C, const float* medians, int32_t* sym,
                   int32_t* idx, hipStream_t s, int perm = 0);
int launch_z_dequant(const int32_t* sym, int B, int h, int w, int C, const float* medians, float* zhat, int zcs,
                     hipStream_t s, int perm = 0);

// One wave per stream.  counts[s] symbols starting at sym_base[s]; writes words backwards into
// out + s*cap_words; out_words[s] receives the number of 32-bit words produced (stream = last out_words words).
// Streams [0, split) use tables t0, streams [split, nstreams) use t1 (rgb / depth in one launch).
int launch_rans_encode(const int32_t* sym, const int32_t* idx, const int64_t* sym_base, const int64_t* counts,
                       int nstreams, int split, DevTables t0, DevTables t1, uint32_t* out, int64_t cap_words,
                       int64_t* out_words, int* err, hipStream_t s);
// Stateful decode: state[s] = {x, pos}; init=1 loads the state from the first two words of each stream.
int launch_rans_decode(const uint32_t* streams, const int64_t* stream_off_words, const int64_t* stream_len_words,
                       int nstreams, uint64_t* state, int init, const int32_t* idx, int32_t* sym,
                       const int64_t* sym_base, int64_t part_off, int64_t count, DevTables t, hipStream_t s);
