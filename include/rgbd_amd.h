/*
 * rgbd_amd.h -- C ABI of the MI355X-native ELIC_united encode/decode path.
 *
 * Shared library: learning-based-rgb-d-image-compression_amd/librgbd_amd.so (built by __graft_entry__.build()).
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative errno-style code
 * (-22 invalid argument, -12 out of memory, -28 buffer too small, -5 HIP runtime error, -1 wrong call order);
 * no exceptions cross the boundary; a handle may be used from one thread at a time.  "dev" pointers are HIP device
 * pointers (e.g. torch tensor .data_ptr() on ROCm); all others are host pointers.  `stream` is a hipStream_t passed
 * as void* (NULL = default stream).
 *
 * Each entry point names the reference interface it stands in for (paths relative to the reference repository).
 */
#ifndef RGBD_AMD_H
#define RGBD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGBD_AMD_ABI_VERSION 1

int rgbd_abi_version(void);

/* Host-side wait policy of the current HIP device: 1 = host threads that wait for the GPU sleep (blocking sync) instead
 * of spinning.  No reference counterpart (the reference drives one image at a time from one thread); a pooled rank
 * keeps 16 host threads waiting on 16 streams, and 8 ranks share one host. */
int rgbd_set_blocking_sync(int32_t on);
/* The policy in force on the current device: 1 blocking sync, 0 anything else, negative on error (tests: CodecPool.close()
 * must give the device its default policy back). */
int rgbd_get_blocking_sync(void);

/* ---------------------------------------------------------------------------------------------------------------
 * Table construction (host, one-off).
 * Replaces compressai._CXX.pmf_to_quantized_cdf -- CompressAI/compressai/cpp_exts/ops/ops.cpp:24-81, bound at
 * ops.cpp:83-90 and called from entropy_models.py:60-63.   cdf_out receives n+1 entries.
 * ------------------------------------------------------------------------------------------------------------- */
int rgbd_pmf_to_quantized_cdf(const float* pmf, int32_t n, int32_t precision, uint32_t* cdf_out);

/* ---------------------------------------------------------------------------------------------------------------
 * Stand-alone rANS coder on the GPU (host buffers in / out).
 * Replaces compressai.ans.{RansEncoder,BufferedRansEncoder,RansDecoder} --
 * CompressAI/compressai/cpp_exts/rans/rans_interface.cpp:99-205 (encode_with_indexes + flush),
 * :207-276 (decode_with_indexes), :278-351 (set_stream / decode_stream), bound at :353-373.
 * cdf is row-major [n_cdf][cdf_stride] int32; cdf_sizes / offsets have n_cdf entries.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct rgbd_tables rgbd_tables; /* packed CDF rows + search accelerator resident in HBM */

int rgbd_tables_create(const int32_t* cdf, int32_t cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets,
                       int32_t n_cdf, rgbd_tables** out);
void rgbd_tables_destroy(rgbd_tables* t);

/* Worst-case stream size in bytes for n symbols (escape-heavy input). */
int64_t rgbd_rans_max_bytes(int64_t n);

/* One stream from n (symbol, index) pairs.  out_len receives the byte count (multiple of 4, >= 8). */
int rgbd_rans_encode(const rgbd_tables* t, const int32_t* symbols, const int32_t* indexes, int64_t n, uint8_t* out,
                     int64_t cap, int64_t* out_len);

typedef struct rgbd_rans_decoder rgbd_rans_decoder;
int rgbd_rans_decoder_create(rgbd_rans_decoder** out);
int rgbd_rans_decoder_set_stream(rgbd_rans_decoder* d, const uint8_t* stream, int64_t nbytes);
/* Decodes n symbols for the given table indexes, continuing from the decoder's current state. */
int rgbd_rans_decoder_decode(rgbd_rans_decoder* d, const rgbd_tables* t, const int32_t* indexes, int64_t n,
                             int32_t* symbols_out);
void rgbd_rans_decoder_destroy(rgbd_rans_decoder* d);

/* The same coder with everything resident in HBM and many streams per launch -- what replaces a Python loop of
 * encode_with_indexes() / decode_stream() calls over images and modalities (elic_united.py:374-401, 543-578;
 * rans_interface.cpp:99-192, 286-351).  Every pointer is a DEVICE pointer; nothing is copied or allocated; the launch is
 * asynchronous on `stream` (a hipStream_t; NULL = the default stream).  Indexes must lie in [0, n_cdf): the codec's own
 * producer (rgbd_ckbd_quant_index) guarantees it, the kernels do not check.
 *   encode: stream s codes counts_dev[s] (symbol, index) pairs starting at element sym_base_dev[s] of symbols_dev / indexes_dev
 *           (both readable one element past the last pair).  out_dev holds nstreams slots of cap_words 32-bit words
 *           (cap_words a multiple of 64, >= rgbd_rans_max_bytes(max count) / 4); stream s is the LAST out_words_dev[s] words of
 *           slot s, byte for byte what RansEncoder.flush() returns.  *err_dev (zero it first) becomes non-zero if a slot
 *           overflowed.
 *   decode: stream s is stream_len_words_dev[s] words at word stream_off_words_dev[s] of streams_dev.  state_dev keeps two
 *           uint64 per stream between calls; init != 0 starts from the head of each stream (set_stream()), init == 0
 *           continues (the next decode_stream() on the same decoder).  Each call decodes `count` symbols per stream for the
 *           indexes at indexes_dev[sym_base_dev[s] + part_off ...) into symbols_dev at the same positions. */
int rgbd_rans_encode_batch_dev(const rgbd_tables* t, const int32_t* symbols_dev, const int32_t* indexes_dev,
                               const int64_t* sym_base_dev, const int64_t* counts_dev, int32_t nstreams, uint32_t* out_dev,
                               int64_t cap_words, int64_t* out_words_dev, int32_t* err_dev, void* stream);
int rgbd_rans_decode_batch_dev(const rgbd_tables* t, const uint32_t* streams_dev, const int64_t* stream_off_words_dev,
                               const int64_t* stream_len_words_dev, int32_t nstreams, uint64_t* state_dev, int32_t init,
                               const int32_t* indexes_dev, int32_t* symbols_dev, const int64_t* sym_base_dev, int64_t part_off,
                               int64_t count, void* stream);

/* One checkerboard half of one channel slice on the encoder: utils/ckbd.py:83-105 (ckbd_anchor_sequeeze /
 * ckbd_nonanchor_sequeeze of y, means and scales), entropy_models.py:118-146 (quantize(y, "symbols", means)), :561-568
 * (build_indexes(scales)) and the scatter of y_hat = symbol + mean back onto the full grid (ckbd.py:107-125), as ONE pass.
 * y_dev / means_dev / scales_dev / yhat_dev: NCHW fp32 [n][c][h][w] on the device, w even.  anchor != 0: the positions with
 * (row + col) odd (ckbd.py:37-48), and the other half of yhat_dev is set to zero; anchor == 0: the positions with (row + col)
 * even, the other half of yhat_dev is left as it is.  scale_table: the 64 entries of get_scale_table() (host).
 * symbols_dev / indexes_dev: n * c * h * (w / 2) int32 each, in (n, c, h, w / 2) order -- the order the reference's
 * .reshape(-1).tolist() feeds its encoder.  One kernel, in place on the caller's tensors, asynchronous on `stream` (a hipStream_t;
 * NULL = the default stream) like the coder entry points above; 8-byte aligned tensors take the vectorised form.
 * rgbd_ckbd_dequant is the decoder's half of the same step (elic_united.py:497-506, 529-538): yhat = symbol + mean. */
int rgbd_ckbd_quant_index(const float* y_dev, const float* means_dev, const float* scales_dev, int32_t n, int32_t c, int32_t h,
                          int32_t w, int32_t anchor, const float* scale_table, int32_t* symbols_dev, int32_t* indexes_dev,
                          float* yhat_dev, void* stream);
int rgbd_ckbd_dequant(const int32_t* symbols_dev, const float* means_dev, int32_t n, int32_t c, int32_t h, int32_t w, int32_t anchor,
                      float* yhat_dev, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Single operators on device tensors (NCHW fp32, contiguous) -- used by the parity tests of the conv kernels.
 * Replaces torch.nn.functional.conv2d / conv_transpose2d as used by modules/layers/conv.py:7-34.
 * weight: Conv2d layout (Cout,Cin,k,k) or ConvTranspose2d layout (Cin,Cout,k,k) when transposed != 0 (host ptr).
 * act: 0 none, 1 ReLU, 2 LeakyReLU(0.01), 3 sigmoid.  residual_dev (optional, output-shaped) is added before act.
 * ------------------------------------------------------------------------------------------------------------- */
int rgbd_conv2d_nchw(const float* x_dev, int32_t n, int32_t cin, int32_t h, int32_t w, const float* weight,
                     const float* bias, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t transposed,
                     int32_t act, const float* residual_dev, float* y_dev, void* stream);

/* The same operator in the arithmetic of the CPU kernels the reference's conv2d / conv_transpose2d calls end in (torch CPU ->
 * oneDNN 3.7.1 jit:avx512_core / jit_1x1:avx512_core, third-party to the reference): per output element a fresh
 * fp32 fma chain per block of input channels (tap-major inside a block, channels ascending), the block sums added in order.
 * blocks: channels per block (nblocks entries; NULL = one block per 16 channels, the multi-tap kernels' structure).
 * bias_mode: 0 = sum, then + bias; 1 = S_0 + bias, then the other block sums; 2 = the first chain starts from the bias.
 * flags bit 0: sigmoid (act 3) as torch's vectorised CPU kernel computes it; bit 1: run the blocks as split-K ranges. */
int rgbd_conv2d_ref_nchw(const float* x_dev, int32_t n, int32_t cin, int32_t h, int32_t w, const float* weight,
                         const float* bias, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t transposed,
                         int32_t act, const float* residual_dev, float* y_dev, void* stream, const int32_t* blocks,
                         int32_t nblocks, int32_t bias_mode, int32_t flags);

/* ---------------------------------------------------------------------------------------------------------------
 * The codec.  Replaces models/elic_united.py: ELIC_united.__init__ :14-86, load_state_dict :588-620,
 * update :580-586, compress :403-427 (+ compress_united :350-401, compress_one_slice :265-348),
 * decompress :429-452 (+ decompress_united :543-578, decompress_one_slice :454-541).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct rgbd_elic rgbd_elic;

/* Reference arithmetic.  An engine created for ELIC_united / ELIC / ELIC_united_R2D computes every float that
 * feeds a coding decision in the accumulation order of the CPU kernels the reference runs on (torch CPU 2.10 -> oneDNN 3.7.1 /
 * MKL / Sleef; third-party to the reference).  Where that order depends on the layer shape -- the reduce blocks of oneDNN's 1x1
 * convolution kernel (kind 0) -- it is data measured on the reference machine and handed over here, per
 * (cin, cout, input h, w, batch of the reference call): blocks[0..nblocks) = channels per block.  Shapes without an entry
 * run as a single block.  rgbd_elic_get_refnum: 1 when the engine uses this arithmetic (STF_united: 0). */
int rgbd_elic_set_ref_blocks(rgbd_elic* m, int32_t kind, int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t batch,
                             const int32_t* blocks, int32_t nblocks);
int rgbd_elic_get_refnum(const rgbd_elic* m);
int rgbd_elic_ref_table_misses(const rgbd_elic* m);

/* A second instance that borrows `src`'s packed device weights and tables (own workspace, own stream): several
 * instances on one GPU then cost one weight copy.  `src` must outlive the clone. */
int rgbd_elic_clone_shared(const rgbd_elic* src, rgbd_elic** out);

/* config/config.py:5-10: N, M and the channel count of each latent slice. */
int rgbd_elic_create(int32_t N, int32_t M, const int32_t* slice_ch, int32_t n_slices, rgbd_elic** out);
void rgbd_elic_destroy(rgbd_elic* m);

/* One state_dict entry (float32, host, reference name and shape).  Call for every parameter, then finalize. */
int rgbd_elic_set_tensor(rgbd_elic* m, const char* name, const float* data, const int64_t* shape, int32_t ndim);
/* which: 0 rgb gaussian, 1 depth gaussian, 2 rgb bottleneck, 3 depth bottleneck (results of update()). */
int rgbd_elic_set_tables(rgbd_elic* m, int32_t which, const int32_t* cdf, int32_t cdf_stride, const int32_t* cdf_sizes,
                         const int32_t* offsets, int32_t n_cdf);
/* scale_table: 64 floats (utils/moduleFunc.py:11-12). */
int rgbd_elic_set_scale_table(rgbd_elic* m, const float* table, int32_t n);
/* Packs the weights for the MFMA kernels and uploads everything to the current device. */
int rgbd_elic_finalize(rgbd_elic* m);

/*
 * compress(): rgb_dev [B,3,H,W], depth_dev [B,1,H,W] fp32 on the device, H and W multiples of 64.
 * per_image_streams = 0 reproduces the reference's batched call (ONE y-stream per modality for the whole batch,
 * elic_united.py:392-400); 1 emits one y-stream per image (= what B separate reference calls produce).
 * z-streams are always per image.  Results are fetched with rgbd_elic_stream().
 */
int rgbd_elic_compress(rgbd_elic* m, const float* rgb_dev, const float* depth_dev, int32_t B, int32_t H, int32_t W,
                       int32_t per_image_streams, void* stream);
/* modality: 0 rgb, 1 depth; kind: 0 y, 1 z; index < count.  Pointer stays valid until the next compress(). */
int rgbd_elic_stream_count(const rgbd_elic* m, int32_t modality, int32_t kind);
int rgbd_elic_stream(const rgbd_elic* m, int32_t modality, int32_t kind, int32_t index, const uint8_t** data,
                     int64_t* nbytes);

/*
 * decompress(): streams as produced above (y: n_y = 1 or B per modality; z: B per modality), z-grid zh x zw
 * (= H/64, W/64).  Writes x_hat clamped to [0,1]: xr_dev [B,3,64*zh,64*zw], xd_dev [B,1,64*zh,64*zw].
 */
int rgbd_elic_decompress(rgbd_elic* m, const uint8_t* const* y_rgb, const int64_t* y_rgb_len, int32_t n_y,
                         const uint8_t* const* y_depth, const int64_t* y_depth_len, const uint8_t* const* z_rgb,
                         const int64_t* z_rgb_len, const uint8_t* const* z_depth, const int64_t* z_depth_len, int32_t B,
                         int32_t zh, int32_t zw, float* xr_dev, float* xd_dev, void* stream);

/*
 * The Bi-CEE entropy stage alone (BASELINE config 4): replaces ELIC_united.compress_united (models/elic_united.py:350-401,
 * with compress_one_slice :265-348) and decompress_united (:543-578, decompress_one_slice :454-541).  Latents y
 * [B,M,h,w] and hyper parameters [B,2M,h,w] are NCHW fp32 device tensors (w even); only y-streams are produced
 * (fetch them with rgbd_elic_stream(kind = 0)); decompress_united writes y_hat [B,M,h,w] per modality.
 */
int rgbd_elic_compress_united(rgbd_elic* m, const float* y_rgb_dev, const float* hyper_rgb_dev, const float* y_depth_dev,
                              const float* hyper_depth_dev, int32_t B, int32_t h, int32_t w, int32_t per_image_streams,
                              void* stream);
int rgbd_elic_decompress_united(rgbd_elic* m, const uint8_t* const* y_rgb, const int64_t* y_rgb_len, int32_t n_y,
                                const uint8_t* const* y_depth, const int64_t* y_depth_len, const float* hyper_rgb_dev,
                                const float* hyper_depth_dev, int32_t B, int32_t h, int32_t w, float* yhat_rgb_dev,
                                float* yhat_depth_dev, void* stream);

/*
 * Single-modal ELIC (BASELINE config 1): replaces models/elic.py: ELIC.__init__ :15-57, compress :161-253,
 * decompress :255-325 (x_hat is NOT clamped there).  Same life cycle as above (set_tensor with the reference's key names,
 * set_tables which = 0 gaussian / 2 bottleneck, set_scale_table, finalize); streams are fetched with
 * rgbd_elic_stream(modality = 0, kind, index).  x_dev: [B,in_ch,H,W] NCHW fp32, H and W multiples of 64.
 */
int rgbd_elic_create_single(int32_t N, int32_t M, const int32_t* slice_ch, int32_t n_slices, int32_t in_ch, rgbd_elic** out);
int rgbd_elic_compress_single(rgbd_elic* m, const float* x_dev, int32_t B, int32_t H, int32_t W, int32_t per_image_streams,
                              void* stream);
int rgbd_elic_decompress_single(rgbd_elic* m, const uint8_t* const* y, const int64_t* y_len, int32_t n_y,
                                const uint8_t* const* z, const int64_t* z_len, int32_t B, int32_t zh, int32_t zw,
                                float* x_dev, void* stream);
/* Eval-mode forward() of the single-modal model (models/elic.py:60-161 with config quant = "ste"): x_hat [B,in_ch,H,W]
 * (not clamped), likelihoods of y [B,M,H/16,W/16] and of z [B,N,H/64,W/64] ("y_likelihoods" / "z_likelihoods"). */
int rgbd_elic_forward_single(rgbd_elic* m, const float* x_dev, int32_t B, int32_t H, int32_t W, float* xhat_dev, float* lik_y,
                             float* lik_z, void* stream);

/*
 * STF_united (BASELINE config 5): replaces models/stf_united.py: SymmetricalTransFormerUnited :605-678 with
 * AnalysisTransformSTFunited :403-502 / SynthesisTransformSTFunited :505-602 (Swin blocks :118-214, PatchMerging :217-249,
 * PatchSplit :252-267, BasicLayer :270-366, PatchEmbed :369-400).  Everything behind the transforms is ELIC_united's (same
 * compress / decompress / forward entry points above); nn.Linear weights are passed as (out, in, 1, 1).
 */
int rgbd_elic_create_stf(int32_t N, int32_t M, const int32_t* slice_ch, int32_t n_slices, rgbd_elic** out);
/* ELIC_united_R2D: replaces models/elic_united_R2D.py:9-326 (AnalysisTransformEXSingle analysis.py:56-112,
 * SynthesisTransformEXSingle synthesis.py:186-242, HyperSynthesisEXSingle synthesis.py:325-343, the one-directional context
 * wiring of compress_one_slice / decompress_one_slice).  Same entry points as ELIC_united otherwise. */
int rgbd_elic_create_r2d(int32_t N, int32_t M, const int32_t* slice_ch, int32_t n_slices, rgbd_elic** out);

/*
 * Eval-mode forward(): replaces ELIC_united.forward / entropy_estimate_united / codeOnePart (models/elic_united.py:94-263)
 * and the likelihood halves of EntropyBottleneck.forward / GaussianConditional.forward (entropy_models.py:391-428,
 * 534-558).  x_hat is NOT clamped (as in the reference); likelihoods are lower-bounded at 1e-9.
 * Outputs (device, NCHW fp32): xr [B,3,H,W], xd [B,1,H,W], lik_y_* [B,M,H/16,W/16], lik_z_* [B,N,H/64,W/64].
 */
int rgbd_elic_forward(rgbd_elic* m, const float* rgb_dev, const float* depth_dev, int32_t B, int32_t H, int32_t W,
                      float* xr_dev, float* xd_dev, float* lik_y_rgb, float* lik_y_depth, float* lik_z_rgb,
                      float* lik_z_depth, void* stream);

/* Intermediate of the last compress()/decompress() as NCHW fp32 on the host (parity tests).  Names: y_r y_d z_r z_d
 * zhat_r zhat_d hyper_r hyper_d yhat_r yhat_d.  shape_out receives 4 ints; data may be NULL to query the shape. */
int rgbd_elic_debug_tensor(rgbd_elic* m, const char* name, float* data, int64_t cap_floats, int32_t* shape_out);
/* Symbols / indexes of the last compress() in stream order (modality 0/1), total count in *n. */
int rgbd_elic_debug_symbols(rgbd_elic* m, int32_t modality, int32_t* symbols, int32_t* indexes, int64_t cap, int64_t* n);
/* Parity bookkeeping: with set_debug_floats(1) the next compress() also keeps, per symbol and in stream order, the value
 * it rounded (y - mean: the argument of quantize(), entropy_models.py:131-137) and the scale it indexed (build_indexes(),
 * entropy_models.py:561-568); debug_floats copies them out.  Lets a test put the GPU's floats next to the reference's at
 * the first symbol where the two streams part (tests/test_gpu_parity_pinned.py). */
int rgbd_elic_set_debug_floats(rgbd_elic* m, int32_t on);
int rgbd_elic_debug_floats(rgbd_elic* m, int32_t modality, float* x, float* scale, int64_t cap, int64_t* n);
/* Teacher forcing (parity bookkeeping; tests/test_gpu_parity_pinned.py::test_teacher_forced_*): the following compress() /
 * compress_united() calls still take every decision from their own floats -- debug_symbols / debug_floats / the streams
 * are the GPU's -- but the context later parts see is rebuilt from the symbols given here, in stream order (what
 * models/elic_united.py:265-348 would have fed forward had it taken exactly these decisions): z_hat = z_sym + median
 * after the z stage (entropy_models.py:437-446), y_hat = y_sym + mean after each of the 20 coding parts.  With the
 * reference's symbols (tests/golden/margins_*.npz) every part BEHIND a first flip is compared under the reference's
 * context.  n_y = B * M * h * w per modality (n_z = B * N * zh * zw; 0 = this stage is not forced); n_y = n_z = 0 clears. */
int rgbd_elic_set_forced_symbols(rgbd_elic* m, int32_t modality, const int32_t* y_sym, int64_t n_y, const int32_t* z_sym,
                                 int64_t n_z);

/* Test hooks: force a split-K factor for rgbd_conv2d_nchw / the codec's entropy-model layers (0 = automatic) and
 * kernel-only timing of one convolution shape on NHWC scratch buffers (tools/conv_sweep.py). */
int rgbd_debug_force_splitk(int32_t s);
/* ResidualBottleneck / ResidualUnit tails (3x3 + ReLU -> 1x1 + residual; res_blk.py:7-27, layers.py:177-196) run as one
 * launch where that is faster, together with the leading 1x1 + ReLU of the block that follows.  -1 = automatic (default),
 * 0 = never, 1 / 2 / 4 = always, with 64 / 128 / 256-pixel tiles; + 16 (15, 17, 18, 20) = the same without the following
 * block's leading layer.  Results are bit-identical in every mode. */
int rgbd_debug_force_fuse(int32_t mode);
/* The image-producing ConvTranspose2d (N -> 3 / 1, k 5, stride 2; synthesis.py:147,168) runs as one 9-tap sub-pixel conv
 * over the input grid (16 channels = 4 output phases x 4) instead of four phases with the couts padded to 16 each:
 * 0 = per-phase form, 1 = sub-pixel form inside the codec (default), 2 = also in rgbd_conv2d_nchw.  Same bits.
 * Mode 0 also returns the image-consuming first conv (3 / 1 -> N, k 5, stride 2; analysis.py:125,150) from its K-packed
 * 1x1 form (25 taps x C real inputs gathered into 80 / 32 channels) to the tap-by-tap form. */
int rgbd_debug_force_subpix(int32_t mode);
/* ELIC_united's RGB and depth branches run the same layer shapes on independent data (analysis.py:116-174,
 * synthesis.py:126-184,305-323, the per-slice channel-context nets of elic_united.py:288-333): 1 (default) issues each such
 * layer pair as ONE grouped launch (twice the workgroups, half the launches), 0 issues two launches.  Same bits either way. */
int rgbd_debug_force_pair(int32_t mode);
/* Convolution tile tables: mode 0 (default) = the winners of isolated launches (lowest latency of one compress / decompress),
 * mode 1 = the winners with the chip shared between several engine instances (highest job throughput; CodecPool sets it).
 * Results are bit-identical in both modes -- tile choice never changes an output. */
int rgbd_elic_set_tile_mode(rgbd_elic* m, int32_t mode);
int rgbd_debug_bench_streams(int32_t n); /* rgbd_conv_bench: issue every launch on n streams at once (1 = isolated) and
                                            report the time per launch -- the cost of a launch on a shared chip */
int rgbd_debug_force_blocked(int32_t on); /* rgbd_conv_bench: time the blocked-accumulation kernels (tools/tune_tiles.py --blocked) */
int rgbd_debug_force_ckbd(int32_t part); /* rgbd_conv2d_nchw / rgbd_conv_bench: 0 = all outputs, 1 = anchor positions only
                                           ((row + col) odd, utils/ckbd.py:37-48), 2 = non-anchor only; the rest reads 0 */
int rgbd_debug_conv_log(int32_t on);                      /* record the shape of every conv launch (tools/tune_tiles.py) */
int64_t rgbd_debug_conv_log_read(char* buf, int64_t cap); /* CSV text of the recorded shapes; returns the size needed */
int rgbd_debug_force_tile(const char* cfg); /* "wm,mt,nt,kc,dma" or "" = automatic (tools/tile_sweep.py) */
/* In-situ tuning (tools/tune_insitu.py): tile / staging form per layer-shape key, lines of
 * "N,H,W,cin_pad,cout_pad,ntaps,stride,nphase,splitk,wm,mt,nt,kc,dma"; "" clears.  rgbd_elic_set_profile(m, 2) makes the profile's
 * layer names carry the shape key of every launch, so one codec call times every layer under its candidate. */
int rgbd_debug_tile_override(const char* csv);

/* The pointwise operators of Bi-SPF / ESA / SE_Block alone (test hook; NCHW device tensors in and out, host weights):
 * op 0 = F.max_pool2d(kernel 7, stride 3) (attention.py:87), 1 = F.interpolate(bilinear, align_corners=False) to (oh, ow)
 * (attention.py:91), 2 = SE_Block x * gate (attention.py:52-67; w0 = fc.0.weight [c/16][c], w1 = fc.2.weight [c][c/16]),
 * 3 = x + x * gate as the entropy-parameter nets use it (entropy.py:75). */
int rgbd_pointwise_nchw(int32_t op, const float* x_dev, int32_t n, int32_t c, int32_t h, int32_t w, int32_t oh, int32_t ow,
                        const float* w0, const float* w1, float* y_dev, void* stream);

/* Kernel-only timing of one convolution shape on NHWC scratch buffers (tools/conv_sweep.py, tools/tune_tiles.py): iters launches,
 * *ms_out = milliseconds per launch. */
int rgbd_conv_bench(int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t k, int32_t stride, int32_t pad,
                    int32_t transposed, int32_t with_residual, int32_t iters, float* ms_out);
int rgbd_elic_profile_dump(rgbd_elic* m, const char* path);

/* Measurement hook (bench.py): when on, every convolution launch is bracketed by HIP events on the launch stream.
 * profile_read returns the summed kernel time (ms), the launch count and the algorithmic FLOPs (2*MACs, unpadded)
 * accumulated since set_profile(). */
/* ---------------------------------------------------------------------------------------------------------------
 * Harness metric: MS-SSIM statistics on the GPU.
 * Replaces the pytorch_msssim.ms_ssim call of utils/metrics.py:8-14 (testing/tester_united.py:92-96) -- 11-tap Gaussian
 * (sigma 1.5) "valid" filtering, SSIM and contrast-structure maps, five dyadic scales with 2x2 average pooling.
 * x, y: device [P][H][W] fp32 planes (P = N * C, contiguous); out: device [P][5][2] = mean SSIM / mean CS per plane and
 * scale, combined by the caller (relu, the five weights, product over scales, mean over channels: rgbd_amd/metrics.py).
 * taps11: the 11 filter taps (host); clamp01: clamp both inputs to [0, 1] first (metrics.py:9-10).  min(H, W) > 160.
 * workspace: device scratch of at least rgbd_msssim_workspace_bytes(P, H, W) bytes.  Deterministic (no atomics).
 * ------------------------------------------------------------------------------------------------------------- */
int64_t rgbd_msssim_workspace_bytes(int32_t P, int32_t H, int32_t W);
int rgbd_msssim_stats(const float* x, const float* y, int32_t P, int32_t H, int32_t W, const float* taps11, float data_range,
                      int32_t clamp01, float* out, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * STF_united operator: nn.LayerNorm(C, eps = 1e-5) over the channels of a token (models/stf_united.py:143,155,225,263,
 * 387-391).  x: device [ntok][xcs] fp32 (the first C of xcs channels are the token), w / b: device [C], y: device
 * [ntok][ycs] (channels C .. ycs - 1 are zeroed).  Biased variance, two passes, fixed summation tree.
 * rgbd_debug_force_layernorm_form: -1 by shape (default), 0 one wavefront per token, 1 sixteen lanes per token (C % 4 == 0)
 * -- the two forms are bit-identical (tests).
 * ------------------------------------------------------------------------------------------------------------- */
int rgbd_layernorm(const float* x, int64_t ntok, int32_t C, int32_t xcs, const float* w, const float* b, float* y, int32_t ycs,
                   void* stream);
void rgbd_debug_force_layernorm_form(int32_t form);

/* Bytes of HBM workspace this engine instance holds (grows with the largest call shape seen, never shrinks); the packed
 * weights, shared by all instances of a pool, are not included.  bench.py reports it as config.hbm_workspace_gib. */
int64_t rgbd_elic_workspace_bytes(const rgbd_elic* m);

/* Number of call shapes whose kernel sequence is currently cached as a HIP graph (tests / diagnostics). */
int rgbd_elic_graph_count(const rgbd_elic* m);
/* Test hook: the next n graph captures are treated as lost (as if another library's device-wide call had invalidated
 * them).  The affected calls must still return correct results (they re-run eagerly); an entry that loses three captures
 * keeps launching eagerly. */
int rgbd_debug_fail_captures(int32_t n);
int rgbd_elic_set_profile(rgbd_elic* m, int32_t on);
int rgbd_elic_profile_read(rgbd_elic* m, double* conv_ms, int64_t* launches, double* flops);
/* `flops` above are ALGORITHMIC: the FLOPs of the reference's layers (what bench.py's roofline divides by time).  A launch that
 * computes one checkerboard half of a layer's outputs (the entropy-parameter nets' last layer, utils/ckbd.py:83-125) or meets
 * only half of the taps with non-zero inputs (local-context convs on an anchor-only slice) EXECUTES fewer: this is their sum,
 * so that executed FLOP/s -- what the MFMA-busy counter sees -- can be reported beside the algorithmic figure. */
int rgbd_elic_profile_read_executed(rgbd_elic* m, double* flops_executed);

#ifdef __cplusplus
}
#endif
#endif /* RGBD_AMD_H */
