// Implicit-GEMM convolution / transposed convolution for gfx950 (MI355X), fp32 in / fp32 accumulate on the
// matrix cores (v_mfma_f32_16x16x4_f32).  Replaces every nn.Conv2d / nn.ConvTranspose2d of the reference's
// ELIC_united path (modules/layers/conv.py:7-34 and the convs inside res_blk.py, attention.py, context.py,
// entropy.py, compressai/layers/layers.py) -- see DESIGN.md "K1/K2".
//
// Data layout: activations NHWC with a channel stride that is a multiple of 16 (pad channels are zero), weights
// pre-packed as [cout_pad][tap][cin_pad].  GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel] with
// k = (ci-chunk of 16, tap, ci).  One 256-thread workgroup owns TM couts x TP pixels (a TH x TW spatial tile of one
// image).  Per 16-channel chunk the input patch (tile + halo) is staged ONCE in LDS and every tap reads it at a
// shifted address, so global traffic per workgroup is patch + weights instead of taps x tile.
//
// Numerics: every output element is one k-ordered fp32 fma chain (MFMA f32 semantics) whose order depends only on
// the layer (chunk -> tap -> channel), never on tiling or batch size: results are run-to-run, batch- and
// tile-invariant, which the entropy decoder relies on to reproduce the encoder's means/scales bit for bit.
#include <array>
#include <map>
#include <string>

#include "conv_mfma_body.h"

// conv_mfma_blk.hip: the blocked-accumulation instantiations (ConvArgs::blocked)
bool conv_blk_tile_ok(int wm, int mt, int nt, int max_tiles);
int launch_conv_blk(const ConvArgs& a, const Choice& c, hipStream_t s);
int launch_conv_fused_blk(const ConvArgs& a, int cls, hipStream_t s);

namespace {

Choice choose(const ConvArgs& a, int max_tiles = 16)
{
    // Pick the tile by a small cost model (cycles per CU): MFMA time of the workgroups a CU has to run vs the bytes
    // each workgroup pulls from L2 (its weight slab K*TM plus the input patch per channel chunk).  Multi-tap layers on
    // small feature maps are weight-traffic bound, so narrow cout tiles x wide pixel tiles win there; big feature maps
    // pick the large square-ish tiles.  The choice never changes results: every output keeps its fma chain.
    static const int cand[][3] = {{2, 3, 8}, {2, 2, 8}, {2, 5, 4}, {2, 4, 4}, {2, 3, 4}, {2, 2, 4}, {2, 1, 4}, {2, 5, 2}, {2, 4, 2},
                                  {2, 3, 2}, {2, 2, 2}, {2, 1, 2}, {2, 5, 1}, {2, 4, 1}, {2, 3, 1}, {2, 2, 1},
                                  {2, 1, 1}, {1, 3, 2}, {1, 2, 2}, {1, 1, 2}, {1, 3, 1}, {1, 2, 1}, {1, 1, 1}};
    int max_taps = 1;
    for (int p = 0; p < a.nphase; ++p) max_taps = a.taps.n[p] > max_taps ? a.taps.n[p] : max_taps;
    const double K = (double)max_taps * a.cin_pad / a.splitk;
    const int ck = a.ckbd ? 1 : 0;
    const int GWe = ck ? (a.GW + 1) / 2 : a.GW;  // computed columns per row
    Choice best{2, 4, 4, 16, 4, false};
    double best_cost = -1.0;
    long best_blocks = 0;
    for (const auto& c : cand) {
        const int wm = c[0], mt = c[1], nt = c[2];
        if (a.blocked && !conv_blk_tile_ok(wm, mt, nt, ring_ok(a) ? 16 : max_tiles)) continue;  // (two accumulator sets per tile)
        const int tm = 16 * mt * wm, tp = 16 * nt * (wm == 2 ? 2 : 4);
        const int twl = pick_tw_log2(GWe, a.GH, tp);
        const int TW = 1 << twl, TH = tp / TW;
        const long tiles = (long)((GWe + TW - 1) / TW) * ((a.GH + TH - 1) / TH) * a.N * (a.groups == 2 ? 2 : 1);
        const long blocks = tiles * ((a.cout_pad + tm - 1) / tm) * a.nphase * a.splitk;
        const int PH = (TH - 1) * a.IS + a.span_y, PW = ((TW << ck) - 1) * a.IS + a.span_x;
        const int pr = tp >= 128 ? 12 : (tp >= 64 ? 6 : 4);
        if ((long)PH * PW * 4 > pr * 256 || (long)tm * 4 > 8 * 256) continue;  // register-staging limits (KC=16)
        // a wave with one or two accumulator tiles cannot cover the MFMA latency / its LDS reads
        const double ilp = mt * nt >= 4 ? 1.0 : (mt * nt >= 2 ? 0.7 : 0.45);
        const double mfma = (double)mt * nt * (K / 4.0) * 32.0 / ilp + 3000.0;      // cycles per workgroup (+ prologue)
        const double load = (K * tm * 4.0 + (double)PH * PW * a.cin_pad * 4.0) / 12.0;  // ~12 B/clk/CU from L2
        const long nb = (blocks + 255) / 256;
        double cost = nb >= 2 ? nb * (mfma > load ? mfma : load) * (blocks < 512 ? 1.15 : 1.0) : (mfma + load);
        // 256-pixel tiles stage half the weights per MFMA: measured +5-8 % once they still fill every CU twice, a loss below
        if (nt == 8) cost *= blocks >= 512 ? 0.93 : 1.25;
        if (best_cost < 0 || cost < best_cost * 0.999) {
            best_cost = cost;
            best = Choice{wm, mt, nt, 16, twl, false};
            best_blocks = blocks;
        }
    }
    // channels per stage: 64 when one stage can hold every tap (keeps the canonical order) and a 16-channel stage
    // would hold too few MFMAs between barriers
    if (a.cin_pad >= 64) {
        const int tp = 16 * best.nt * (best.wm == 2 ? 2 : 4);
        const int TW = 1 << best.tw_log2, TH = tp / TW;
        const size_t patch64 = (size_t)((TH - 1) * a.IS + a.span_y) * (((TW << ck) - 1) * a.IS + a.span_x) * 68 * 4;
        const size_t tap64 = (size_t)16 * best.mt * best.wm * 68 * 4;
        const long mfma_per_stage16 = (long)max_taps * 4 * best.mt * best.nt;
        const long p4 = (long)((TH - 1) * a.IS + a.span_y) * (((TW << ck) - 1) * a.IS + a.span_x) * 16;
        const long w4 = (long)max_taps * 16 * best.mt * best.wm * 16;
        const int pr = tp >= 128 ? 12 : (tp >= 64 ? 6 : 4);
        if (patch64 + (size_t)max_taps * tap64 <= (size_t)LDS_BUDGET && mfma_per_stage16 < 400 && p4 <= pr * 256 &&
            w4 <= 8 * 256)
            best.kc = 64;
    }
    // direct-to-LDS double buffering whenever two patch images and at least two taps of weights fit in the budget
    {
        const int tp = 16 * best.nt * (best.wm == 2 ? 2 : 4);
        const int TW = 1 << best.tw_log2, TH = tp / TW;
        const long patch = (long)((TH - 1) * a.IS + a.span_y) * (((TW << ck) - 1) * a.IS + a.span_x) * 64;
        const long tap = (long)16 * best.mt * best.wm * 64;
        const int need = max_taps < 2 ? max_taps : 2;
        best.dma = best.kc == 16 && 2 * patch + 2 * need * tap <= (long)LDS_BUDGET;
    }
    // 1x1 layers: a ring of four 16-channel stages (three in flight) whenever the tile has whole waves of patch slots
    {
        static const bool ring_off = getenv("RGBD_NO_RING") != nullptr;
        const int tm = 16 * best.mt * best.wm, tp = 16 * best.nt * (best.wm == 2 ? 2 : 4);
        if (!ring_off && ring_ok(a) && tp % 64 == 0) {
            const long stage = (long)(tm + tp) * 64;
            if (4 * stage + 256 <= (long)LDS_BUDGET + 2048) set_mode(best, 16, 4);
            else if (tp == 256) set_mode(best, 16, 5);
        }
    }
    (void)best_blocks;
    // blocked accumulation: 12 result tiles per wave fit two workgroups per CU only in the direct-to-LDS form (no staging
    // registers), 16 only in the ring form (conv_mfma_blk.hip)
    if (a.blocked && !best.ring && best.mt * best.nt > 10) {
        if (best.mt * best.nt > 12 || !best.dma) {
            if (max_tiles > 10) return choose(a, best.mt * best.nt > 12 ? 12 : 10);
        }
    }
    return best;
}

// ring staging: every tile with whole waves of patch slots (TP % 64 == 0); the 256-pixel tiles also with three buffers
#define RGBD_RING4(WM_, WN_, MT_, NT_)                                             \
    if (c.ring == 4 && c.wm == WM_ && c.mt == MT_ && c.nt == NT_)                  \
        return launch_cfg<WM_, WN_, MT_, NT_, 16, true, 0, false, 4>(a, c.tw_log2, s, 160 * 1024);
#define RGBD_RING3(WM_, WN_, MT_, NT_)                                             \
    if (c.ring == 3 && c.wm == WM_ && c.mt == MT_ && c.nt == NT_)                  \
        return launch_cfg<WM_, WN_, MT_, NT_, 16, true, 0, false, 3>(a, c.tw_log2, s, 160 * 1024);

#define RGBD_CASE(WM_, WN_, MT_, NT_)                                              \
    if (c.wm == WM_ && c.mt == MT_ && c.nt == NT_)                                 \
        return c.kc == 64 ? launch_cfg<WM_, WN_, MT_, NT_, 64, false>(a, c.tw_log2, s)            \
                          : (c.dma ? launch_cfg<WM_, WN_, MT_, NT_, 16, true>(a, c.tw_log2, s, c.lds_cap)    \
                                   : launch_cfg<WM_, WN_, MT_, NT_, 16, false>(a, c.tw_log2, s));

}  // namespace

// out = epilogue(P[0] + P[1] + ... in order): the second half of a split-K convolution
__global__ void splitk_reduce_kernel(ConvArgs a, size_t total4)
{
    const int c4n = a.cout_pad / 4;
    const size_t plane = (size_t)a.N * a.OH * a.OW * a.cout_pad;
    const size_t all4 = a.groups == 2 ? 2 * total4 : total4;
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < all4; j += (size_t)gridDim.x * blockDim.x) {
        const bool g1 = j >= total4;  // grouped launch: the second operand set's outputs
        const size_t i = g1 ? j - total4 : j;
        const float* const gpartial = g1 ? a.g1.partial : a.partial;
        const float* const gbias = g1 ? a.g1.bias : a.bias;
        const float* const gres1 = g1 ? a.g1.res1 : a.res1;
        const float* const gmul = g1 ? a.g1.mul : a.mul;
        const float* const gres2 = g1 ? a.g1.res2 : a.res2;
        float* const gy = g1 ? a.g1.y : a.y;
        const size_t pix = i / c4n;
        const int cb = (int)(i - pix * c4n) * 4;
        if (cb >= a.cout_store) continue;
        if (a.ckbd) {  // checkerboard output: the other half was not computed
            const int gx = (int)(pix % a.OW), gy = (int)((pix / a.OW) % a.OH);
            if (((gy + gx) & 1) != (a.ckbd == 1 ? 1 : 0)) continue;
        }
        f32x4 v = *reinterpret_cast<const f32x4*>(gpartial + pix * a.cout_pad + cb);
        for (int s = 1; s < a.splitk; ++s) v += *reinterpret_cast<const f32x4*>(gpartial + s * plane + pix * a.cout_pad + cb);
        if (a.bias_mode == 0) v += *reinterpret_cast<const f32x4*>(gbias + cb);  // (modes 1 / 2: inside the first partial)
        if (a.res1) v += *reinterpret_cast<const f32x4*>(gres1 + pix * a.r1cs + cb);
        if (a.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (a.act == ACT_LEAKY) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * 0.01f;
        } else if (a.act == ACT_SIGMOID) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = a.exact_math ? rgbd_sigmoid_ref(v[e]) : sigmoid_f32(v[e]);
        } else if (a.act == ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
        }
        if (a.mul) v *= *reinterpret_cast<const f32x4*>(gmul + pix * a.mcs + cb);
        if (a.res2) v += *reinterpret_cast<const f32x4*>(gres2 + pix * a.r2cs + cb);
        *reinterpret_cast<f32x4*>(gy + pix * a.ycs + cb) = v;
    }
}

static int launch_conv_main(const ConvArgs& a, hipStream_t s);

// grouped launch: the second operand set mirrors the first, pointer by pointer
static bool conv_groups_ok(const ConvArgs& a)
{
    if (a.groups != 2) return a.groups == 0 || a.groups == 1;
    const ConvPtrs& g = a.g1;
    return g.x && g.w && g.bias && g.y && !a.res1 == !g.res1 && !a.mul == !g.mul && !a.res2 == !g.res2 &&
           !a.partial == !g.partial && !a.y2 == !g.y2 && !a.w2 == !g.w2 && !a.bias2 == !g.bias2 && !a.w3 == !g.w3 &&
           !a.bias3 == !g.bias3 && !a.y3 == !g.y3;
}

char g_conv_force[64] = {0};  // rgbd_debug_force_tile (tools/tile_sweep.py)

// ---- measured tile table -------------------------------------------------------------------------------------------
// tools/tune_tiles.py times every tile shape / stage depth / staging mode for the layer shapes of a workload and writes
// tile_table.h; a launch whose shape is listed takes the measured winner, everything else the cost model.  The choice
// never changes results (every output keeps its fma chain), so the table is a pure performance database.
struct TunedTile {
    int N, H, W, cin_pad, cout_pad, ntaps, stride, nphase, splitk;  // key (nphase + 10 * ckbd for checkerboard launches, + 100 for
                                                                    // the blocked-accumulation kernels)
    int wm, mt, nt, kc, dma;                                        // measured best
};
static const TunedTile kTuned[] = {
#include "tile_table.h"
#include "tile_table_blk.h"
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
// the same shapes timed with 8 copies of the launch in flight (tools/tune_tiles.py --streams 8): what a launch costs in CU
// time on a shared chip.  Engine instances of a pool use it (ConvArgs::loaded); the winners are larger tiles / fewer
// workgroups than the isolated-launch winners (+2.7 % job throughput on c2, but 15 % slower launches on an idle chip).
static const TunedTile kTunedLoaded[] = {
#include "tile_table_loaded.h"
#include "tile_table_blk_loaded.h"
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};

// *measured: some entry of the table has this map size and batch size, i.e. a miss means "the cost model's pick won"
static const TunedTile* table_lookup(const TunedTile* table, const ConvArgs& a, bool* measured)
{
    const int stride = a.nphase > 1 ? a.OS : a.IS;
    const TunedTile* near = nullptr;
    *measured = false;
    const int aN = a.N * (a.groups == 2 ? 2 : 1);  // a grouped launch tiles like the same layer at twice the batch
    for (const TunedTile* t = table; t->N; ++t) {
        if (t->H == a.H && t->W == a.W && t->N == aN) *measured = true;
        if (t->H != a.H || t->W != a.W || t->cin_pad != a.cin_pad || t->cout_pad != a.cout_pad || t->ntaps != a.ntaps_total ||
            t->stride != stride || t->nphase != a.nphase + 10 * a.ckbd + (a.blocked ? 100 : 0) || t->splitk != a.splitk)
            continue;
        if (t->N == aN) return t;
        if (!near || abs(t->N - aN) < abs(near->N - aN)) near = t;
    }
    // A batch size that was never measured on this map takes the entry of the same layer whose batch size is closest --
    // the winner depends on N only through the number of tiles, so a neighbour beats the cost model.
    return *measured ? nullptr : near;
}

static const TunedTile* tuned_lookup(const ConvArgs& a)
{
    static const bool off = getenv("RGBD_NO_TILE_TABLE") != nullptr;
    if (off) return nullptr;
    bool measured = false;
    if (a.loaded) {  // throughput flavour first; maps it never saw fall back to the isolated-launch table
        const TunedTile* t = table_lookup(kTunedLoaded, a, &measured);
        if (t || measured) return t;
    }
    return table_lookup(kTuned, a, &measured);
}

// shape log for the tuner (rgbd_debug_conv_log): key -> launches
static std::mutex g_log_mu;
static bool g_log_on = false;
static std::map<std::string, int> g_log;

int conv_log_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_log_mu);
    g_log_on = on != 0;
    if (on) g_log.clear();
    return RGBD_OK;
}

long conv_log_read(char* buf, long cap)
{
    std::lock_guard<std::mutex> lk(g_log_mu);
    std::string out = "N,H,W,cin_pad,cout_pad,ntaps,stride,nphase,splitk,launches\n";
    for (const auto& kv : g_log) out += kv.first + "," + std::to_string(kv.second) + "\n";
    if (buf && cap > 0) {
        const size_t n = out.size() < (size_t)cap - 1 ? out.size() : (size_t)cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (long)out.size() + 1;
}

// in-situ tile overrides (tools/tune_insitu.py): shape key (as in the conv log) -> tile / staging form.  Consulted before the
// tables; a pure speed matter like every tile choice.  Lines "N,H,W,cin_pad,cout_pad,ntaps,stride,nphase,splitk,wm,mt,nt,kc,dma".
static std::map<std::string, std::array<int, 5>> g_ovr;
int conv_tile_override(const char* csv)
{
    std::lock_guard<std::mutex> lk(g_log_mu);
    g_ovr.clear();
    if (!csv) return RGBD_OK;
    const char* p = csv;
    while (*p) {
        int v[14], n = 0, used = 0;
        while (n < 14 && sscanf(p, "%d%n", &v[n], &used) == 1) {
            p += used;
            ++n;
            if (*p == ',') ++p;
            else break;
        }
        while (*p && *p != '\n') ++p;
        if (*p == '\n') ++p;
        if (n != 14) return RGBD_EINVAL;
        char key[160];
        snprintf(key, sizeof(key), "%d,%d,%d,%d,%d,%d,%d,%d,%d", v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]);
        g_ovr[key] = {v[9], v[10], v[11], v[12], v[13]};
    }
    return RGBD_OK;
}

int launch_conv(const ConvArgs& a_in, hipStream_t s)
{
    ConvArgs a = a_in;
    if (a.cout_store <= 0 || a.cout_store > a.cout_pad) a.cout_store = a.cout_pad;
    if (a.cout_store % 4) return RGBD_EINVAL;
    if (a.splitk < 1) a.splitk = 1;
    if (a.splitk > a.cin_pad / 16) a.splitk = a.cin_pad / 16;
    if (a.splitk > 1 && !a.partial) return RGBD_EINVAL;
    if (!conv_groups_ok(a)) return RGBD_EINVAL;
    {   // the kernels address one image's input / output / fused operands with 32-bit byte offsets from a per-image base
        const size_t lim = (size_t)1 << 32;
        const size_t opx = (size_t)a.OH * a.OW * 4;
        const int ocs = std::max(std::max(a.ycs, a.cout_pad), std::max(a.r1cs, std::max(a.mcs, a.r2cs)));
        if ((size_t)a.H * a.W * a.xcs * 4 >= lim || opx * ocs >= lim ||
            (size_t)a.cout_pad * a.ntaps_total * a.cin_pad * 4 >= lim)
            return RGBD_EINVAL;
    }
    if (a.act == ACT_GELU && !a.partial) return RGBD_EINVAL;  // GELU is applied by the reducer: needs one partial plane
    if (a.y2 && (a.partial || a.subpix || a.y2cs % 4)) return RGBD_EINVAL;  // the reducer / sub-pixel store have one target
    if (a.subpix && (a.cout_pad != 16 || a.nphase != 1 || a.IS != 1 || a.OS != 2 || a.splitk > 1 || a.res1 || a.mul || a.res2 ||
                     a.ckbd || a.ycs < 4))
        return RGBD_EINVAL;
    if (a.ckbd && (a.ckbd > 2 || a.ckbd < 0 || a.nphase != 1 || a.IS != 1 || a.OS != 1)) return RGBD_EINVAL;
    // (Adding the partial planes inside the conv kernel -- the workgroup that finishes a tile last reduces it, found through a
    //  per-tile counter -- was built in round 3, bit-identical, and measured: 69.2 instead of 60.6 ms of conv per c3 step.
    //  Every workgroup of a split launch then pays an agent-scope release / acquire pair, i.e. an L2 write-back and
    //  invalidate across the XCDs, for a reducer launch of a few microseconds.  Not kept)
    const int rc = launch_conv_main(a, s);
    if (rc || !a.partial) return rc;
    const size_t total4 = (size_t)a.N * a.OH * a.OW * (a.cout_pad / 4);
    size_t g = ((a.groups == 2 ? 2 : 1) * total4 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, a, total4);
    HIP_TRY(hipGetLastError());
    return RGBD_OK;
}

static int launch_conv_main(const ConvArgs& a, hipStream_t s)
{
    if (a.cin_pad % 16 || a.cout_pad % 16 || a.xcs % 4 || a.ycs % 4) return RGBD_EINVAL;
    if (a.nphase != 1 && a.nphase != 4) return RGBD_EINVAL;
    if (a.N <= 0 || a.GH <= 0 || a.GW <= 0) return RGBD_EINVAL;
    Choice c = choose(a);
    const TunedTile* t = tuned_lookup(a);
    // (the tables were measured on the single-chain kernels: an entry is taken when its tile exists in blocked form)
    if (t && a.blocked) {
        const int tiles = t->mt * t->nt;
        const bool ring = t->kc == 16 && (t->dma == 4 || t->dma == 5), dma = t->kc == 16 && t->dma != 0;
        if (!conv_blk_tile_ok(t->wm, t->mt, t->nt, 16) || (tiles > 12 && !ring) || (tiles > 10 && !dma)) t = nullptr;
    }
    if (t) {
        c.wm = t->wm;
        c.mt = t->mt;
        c.nt = t->nt;
        set_mode(c, t->kc, t->dma);
        c.tw_log2 = pick_tw_log2(a.ckbd ? (a.GW + 1) / 2 : a.GW, a.GH, 16 * t->nt * (t->wm == 2 ? 2 : 4));
    }
    if (g_log_on) {
        char key[160];
        snprintf(key, sizeof(key), "%d,%d,%d,%d,%d,%d,%d,%d,%d", a.N * (a.groups == 2 ? 2 : 1), a.H, a.W, a.cin_pad, a.cout_pad, a.ntaps_total,
                 a.nphase > 1 ? a.OS : a.IS, a.nphase + 10 * a.ckbd + (a.blocked ? 100 : 0), a.splitk);
        std::lock_guard<std::mutex> lk(g_log_mu);
        ++g_log[key];
    }
    if (!g_ovr.empty()) {
        char key[160];
        snprintf(key, sizeof(key), "%d,%d,%d,%d,%d,%d,%d,%d,%d", a.N * (a.groups == 2 ? 2 : 1), a.H, a.W, a.cin_pad, a.cout_pad, a.ntaps_total,
                 a.nphase > 1 ? a.OS : a.IS, a.nphase + 10 * a.ckbd + (a.blocked ? 100 : 0), a.splitk);
        std::lock_guard<std::mutex> lk(g_log_mu);
        auto it = g_ovr.find(key);
        if (it != g_ovr.end()) {
            c.wm = it->second[0];
            c.mt = it->second[1];
            c.nt = it->second[2];
            set_mode(c, it->second[3], it->second[4]);
            c.tw_log2 = pick_tw_log2(a.ckbd ? (a.GW + 1) / 2 : a.GW, a.GH, 16 * c.nt * (c.wm == 2 ? 2 : 4));
        }
    }
    static const char* force_env = getenv("RGBD_CONV_FORCE");  // "wm,mt,nt[,kc[,dma]]" -- tuning experiments only
    const char* force = g_conv_force[0] ? g_conv_force : force_env;
    if (force) {
        int wm = c.wm, mt = c.mt, nt = c.nt, kc = c.kc, dm = c.dma;
        sscanf(force, "%d,%d,%d,%d,%d", &wm, &mt, &nt, &kc, &dm);
        c.wm = wm;
        c.mt = mt;
        c.nt = nt;
        set_mode(c, kc, dm);
        c.tw_log2 = pick_tw_log2(a.ckbd ? (a.GW + 1) / 2 : a.GW, a.GH, 16 * nt * (wm == 2 ? 2 : 4));
    }
    static const bool debug = getenv("RGBD_CONV_DEBUG") != nullptr;
    if (debug)
        fprintf(stderr, "[conv] N=%d GH=%d GW=%d cin=%d cout=%d taps=%d IS=%d nph=%d -> WM=%d MT=%d NT=%d KC=%d tw=%d dma=%d\n", a.N,
                a.GH, a.GW, a.cin_pad, a.cout_pad, a.taps.n[0], a.IS, a.nphase, c.wm, c.mt, c.nt, c.kc, 1 << c.tw_log2, (int)c.dma);
    // 256-pixel tiles (half the weight staging per MFMA, one round of workgroups on the 128x128 maps): reached through
    // measured table entries only, the cost model does not propose them
    if (c.ring && !ring_ok(a)) return RGBD_ENOSPC;
    if (a.blocked) return launch_conv_blk(a, c, s);
    RGBD_RING4(2, 2, 3, 8) RGBD_RING4(2, 2, 2, 8) RGBD_RING4(2, 2, 1, 8) RGBD_RING4(1, 4, 3, 4) RGBD_RING4(1, 4, 2, 4) RGBD_RING4(1, 4, 1, 4)
    RGBD_RING4(2, 2, 5, 4) RGBD_RING4(2, 2, 4, 4) RGBD_RING4(2, 2, 3, 4) RGBD_RING4(2, 2, 2, 4) RGBD_RING4(2, 2, 1, 4)
    RGBD_RING4(2, 2, 5, 2) RGBD_RING4(2, 2, 4, 2) RGBD_RING4(2, 2, 3, 2) RGBD_RING4(2, 2, 2, 2) RGBD_RING4(2, 2, 1, 2)
    RGBD_RING4(1, 4, 3, 2) RGBD_RING4(1, 4, 2, 2) RGBD_RING4(1, 4, 1, 2) RGBD_RING4(1, 4, 3, 1) RGBD_RING4(1, 4, 2, 1) RGBD_RING4(1, 4, 1, 1)
    // three buffers (two stages in flight, a quarter less LDS: one more workgroup per CU on the small tiles, and the only
    // form of the 256-pixel tiles that leaves room for two)
    RGBD_RING3(2, 2, 3, 8) RGBD_RING3(2, 2, 2, 8) RGBD_RING3(2, 2, 1, 8) RGBD_RING3(1, 4, 3, 4) RGBD_RING3(1, 4, 2, 4) RGBD_RING3(1, 4, 1, 4)
    RGBD_RING3(2, 2, 5, 4) RGBD_RING3(2, 2, 4, 4) RGBD_RING3(2, 2, 3, 4) RGBD_RING3(2, 2, 2, 4) RGBD_RING3(2, 2, 1, 4)
    RGBD_RING3(2, 2, 5, 2) RGBD_RING3(2, 2, 4, 2) RGBD_RING3(2, 2, 3, 2) RGBD_RING3(2, 2, 2, 2) RGBD_RING3(2, 2, 1, 2)
    RGBD_RING3(1, 4, 3, 2) RGBD_RING3(1, 4, 2, 2) RGBD_RING3(1, 4, 1, 2) RGBD_RING3(1, 4, 3, 1) RGBD_RING3(1, 4, 2, 1) RGBD_RING3(1, 4, 1, 1)
    if (c.ring) return RGBD_ENOSPC;
    RGBD_CASE(2, 2, 3, 8) RGBD_CASE(2, 2, 2, 8) RGBD_CASE(2, 2, 1, 8) RGBD_CASE(1, 4, 3, 4) RGBD_CASE(1, 4, 2, 4) RGBD_CASE(1, 4, 1, 4)
    RGBD_CASE(2, 2, 5, 4) RGBD_CASE(2, 2, 4, 4) RGBD_CASE(2, 2, 3, 4) RGBD_CASE(2, 2, 2, 4) RGBD_CASE(2, 2, 1, 4)
    RGBD_CASE(2, 2, 5, 2) RGBD_CASE(2, 2, 4, 2) RGBD_CASE(2, 2, 3, 2) RGBD_CASE(2, 2, 2, 2) RGBD_CASE(2, 2, 1, 2)
    RGBD_CASE(2, 2, 5, 1) RGBD_CASE(2, 2, 4, 1) RGBD_CASE(2, 2, 3, 1) RGBD_CASE(2, 2, 2, 1) RGBD_CASE(2, 2, 1, 1)
    RGBD_CASE(1, 4, 3, 2) RGBD_CASE(1, 4, 2, 2) RGBD_CASE(1, 4, 1, 2)
    RGBD_CASE(1, 4, 3, 1) RGBD_CASE(1, 4, 2, 1) RGBD_CASE(1, 4, 1, 1)
    return RGBD_EINVAL;
}

// ---- conv + fused trailing 1x1 ---------------------------------------------------------------------------------------
// The pair (3x3 C->C, 1x1 C->C2) of a ResidualBottleneck / ResidualUnit as one launch: the first layer's whole cout range
// is one workgroup tile (TM = cout_pad = 96), pixel tiles of 64 / 128 / 256.  Results are bit-identical to the two
// stand-alone launches (tests/test_gpu_conv.py::test_fused_tail_bit_identical), so fusing is a speed decision only.
static const bool g_fuse_off = getenv("RGBD_NO_FUSE") != nullptr;
int g_fuse_force = -1;  // rgbd_debug_force_fuse: -1 = plan, 0 = never, 1 / 2 / 4 = always with that pixel-tile class
int g_fuse_lead_off = 0;  // ... + 16: without the next block's leading 1x1

int conv_fused_plan(int cout_pad, int cout2_pad, int ntaps, int N, int GH, int GW, int loaded)
{
    if (g_fuse_off || g_fuse_force == 0) return 0;
    if (cout_pad != 96 || cout2_pad % 96 || ntaps > 9) return 0;  // instantiated: MT = 6; groups of 2 or 3 cout tiles
    if (g_fuse_force > 0) return g_fuse_force;
    static const char* nt_env = getenv("RGBD_FUSE_NT");
    if (nt_env) return atoi(nt_env);
    const long px = (long)N * GH * GW;
    // enough workgroups to fill 256 CUs twice with the widest tile that still does; a launch that cannot fill the chip
    // once with 64-pixel tiles stays unfused (the stand-alone kernels tile the couts as well)
    if (px / 256 >= (loaded ? 320 : 480)) return 4;
    if (px / 128 >= 320) return 2;
    if (px / 64 >= 256) return 1;
    return 0;
}

int launch_conv_fused(const ConvArgs& a_in, hipStream_t s)
{
    ConvArgs a = a_in;
    if (!a.w2 || !a.bias2 || a.cout2_pad <= 0 || !conv_groups_ok(a)) return RGBD_EINVAL;
    if (a.cout_store <= 0 || a.cout_store > a.cout2_pad) a.cout_store = a.cout2_pad;
    if (a.cout_store % 4) return RGBD_EINVAL;
    if (a.nphase != 1 || a.IS != 1 || a.OS != 1 || a.splitk > 1 || a.partial || a.mul || a.res2 || a.ckbd) return RGBD_EINVAL;
    if (a.act != ACT_NONE && a.act != ACT_RELU && a.act != ACT_LEAKY) return RGBD_EINVAL;
    if (a.act_mid != ACT_NONE && a.act_mid != ACT_RELU) return RGBD_EINVAL;
    if (a.cin_pad % 16 || a.cout_pad % 16 || a.xcs % 4 || a.ycs % 4 || a.N <= 0 || a.GH <= 0 || a.GW <= 0) return RGBD_EINVAL;
    if (a.OH != a.GH || a.OW != a.GW) return RGBD_EINVAL;
    a.splitk = 1;
    {
        const size_t lim = (size_t)1 << 32;
        const size_t opx = (size_t)a.OH * a.OW * 4;
        const int ocs = std::max(std::max(a.ycs, a.cout2_pad), a.r1cs);
        if ((size_t)a.H * a.W * a.xcs * 4 >= lim || opx * ocs >= lim || (size_t)a.cout_pad * a.ntaps_total * a.cin_pad * 4 >= lim)
            return RGBD_EINVAL;
    }
    const int cls = conv_fused_plan(a.cout_pad, a.cout2_pad, a.ntaps_total, a.N * (a.groups == 2 ? 2 : 1), a.GH, a.GW, a.loaded);
    if (a.cout_pad != 96) return RGBD_EINVAL;
    if (a.blocked) return launch_conv_fused_blk(a, cls, s);
    if (a.w3) {  // + the next block's leading 1x1: 128- and 64-pixel tiles only (u needs TM x TP accumulators as well)
        if (a.cout2_pad % 32) return RGBD_EINVAL;
        if (cls >= 2) return launch_cfg<1, 4, 6, 2, 16, true, 2, true>(a, pick_tw_log2(a.GW, a.GH, 128), s);
        if (cls == 1) return launch_cfg<1, 4, 6, 1, 16, true, 2, true>(a, pick_tw_log2(a.GW, a.GH, 64), s);
        return RGBD_EINVAL;
    }
    if (cls == 4) return launch_cfg<1, 4, 6, 4, 16, true, 2>(a, pick_tw_log2(a.GW, a.GH, 256), s);
    if (cls == 2) return launch_cfg<1, 4, 6, 2, 16, true, 3>(a, pick_tw_log2(a.GW, a.GH, 128), s);
    if (cls == 1) return launch_cfg<1, 4, 6, 1, 16, true, 3>(a, pick_tw_log2(a.GW, a.GH, 64), s);
    return RGBD_EINVAL;
}
