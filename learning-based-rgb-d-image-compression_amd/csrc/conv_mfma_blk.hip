// Blocked-accumulation instantiations of the convolution kernels (conv_mfma_body.h, BLK = true) -- the arithmetic of the
// reference's CPU convolutions: per output element one fresh fp32 fma chain per block of input channels
// (16 for the multi-tap layers, the reduce blocks of the 1x1 layers), the block sums added to a running total in channel
// order.  Two accumulator sets per output tile, so only tiles of at most 16 MFMA result tiles per wave are instantiated
// (2 x 64 of the 256 registers a wave has at two workgroups per CU).  Tile choice never changes a result.
#include "conv_mfma_body.h"

namespace {

#define RGBD_BRING4(WM_, WN_, MT_, NT_)                                            \
    if (c.ring == 4 && c.wm == WM_ && c.mt == MT_ && c.nt == NT_)                  \
        return launch_cfg<WM_, WN_, MT_, NT_, 16, true, 0, false, 4, true>(a, c.tw_log2, s, 160 * 1024);
#define RGBD_BRING3(WM_, WN_, MT_, NT_)                                            \
    if (c.ring == 3 && c.wm == WM_ && c.mt == MT_ && c.nt == NT_)                  \
        return launch_cfg<WM_, WN_, MT_, NT_, 16, true, 0, false, 3, true>(a, c.tw_log2, s, 160 * 1024);
#define RGBD_BCASE(WM_, WN_, MT_, NT_)                                                                           \
    if (c.wm == WM_ && c.mt == MT_ && c.nt == NT_)                                                               \
        return c.kc == 64 ? launch_cfg<WM_, WN_, MT_, NT_, 64, false, 0, false, 2, true>(a, c.tw_log2, s)       \
                          : (c.dma ? launch_cfg<WM_, WN_, MT_, NT_, 16, true, 0, false, 2, true>(a, c.tw_log2, s, c.lds_cap) \
                                   : launch_cfg<WM_, WN_, MT_, NT_, 16, false, 0, false, 2, true>(a, c.tw_log2, s));

}  // namespace

bool conv_blk_tile_ok(int wm, int mt, int nt, int max_tiles)
{
    if (mt * nt > 16 || mt * nt > max_tiles) return false;
    if (wm == 2) return (nt == 8 && mt <= 2) || (nt == 4 && mt <= 4) || ((nt == 2 || nt == 1) && mt <= 5);
    return wm == 1 && mt <= 3 && (nt == 4 || nt == 2 || nt == 1);
}

int launch_conv_blk(const ConvArgs& a, const Choice& c, hipStream_t s)
{
    if (!conv_blk_tile_ok(c.wm, c.mt, c.nt, 16) || a.splitk > 1 || a.partial) return RGBD_EINVAL;
    RGBD_BRING4(2, 2, 2, 8) RGBD_BRING4(2, 2, 1, 8) RGBD_BRING4(1, 4, 3, 4) RGBD_BRING4(1, 4, 2, 4) RGBD_BRING4(1, 4, 1, 4)
    RGBD_BRING4(2, 2, 4, 4) RGBD_BRING4(2, 2, 3, 4) RGBD_BRING4(2, 2, 2, 4) RGBD_BRING4(2, 2, 1, 4)
    RGBD_BRING4(2, 2, 5, 2) RGBD_BRING4(2, 2, 4, 2) RGBD_BRING4(2, 2, 3, 2) RGBD_BRING4(2, 2, 2, 2) RGBD_BRING4(2, 2, 1, 2)
    RGBD_BRING4(1, 4, 3, 2) RGBD_BRING4(1, 4, 2, 2) RGBD_BRING4(1, 4, 1, 2) RGBD_BRING4(1, 4, 3, 1) RGBD_BRING4(1, 4, 2, 1) RGBD_BRING4(1, 4, 1, 1)
    RGBD_BRING3(2, 2, 2, 8) RGBD_BRING3(2, 2, 1, 8) RGBD_BRING3(1, 4, 3, 4) RGBD_BRING3(1, 4, 2, 4) RGBD_BRING3(1, 4, 1, 4)
    RGBD_BRING3(2, 2, 4, 4) RGBD_BRING3(2, 2, 3, 4) RGBD_BRING3(2, 2, 2, 4) RGBD_BRING3(2, 2, 1, 4)
    RGBD_BRING3(2, 2, 5, 2) RGBD_BRING3(2, 2, 4, 2) RGBD_BRING3(2, 2, 3, 2) RGBD_BRING3(2, 2, 2, 2) RGBD_BRING3(2, 2, 1, 2)
    RGBD_BRING3(1, 4, 3, 2) RGBD_BRING3(1, 4, 2, 2) RGBD_BRING3(1, 4, 1, 2) RGBD_BRING3(1, 4, 3, 1) RGBD_BRING3(1, 4, 2, 1) RGBD_BRING3(1, 4, 1, 1)
    if (c.ring) return RGBD_ENOSPC;
    RGBD_BCASE(2, 2, 2, 8) RGBD_BCASE(2, 2, 1, 8) RGBD_BCASE(1, 4, 3, 4) RGBD_BCASE(1, 4, 2, 4) RGBD_BCASE(1, 4, 1, 4)
    RGBD_BCASE(2, 2, 4, 4) RGBD_BCASE(2, 2, 3, 4) RGBD_BCASE(2, 2, 2, 4) RGBD_BCASE(2, 2, 1, 4)
    RGBD_BCASE(2, 2, 5, 2) RGBD_BCASE(2, 2, 4, 2) RGBD_BCASE(2, 2, 3, 2) RGBD_BCASE(2, 2, 2, 2) RGBD_BCASE(2, 2, 1, 2)
    RGBD_BCASE(2, 2, 5, 1) RGBD_BCASE(2, 2, 4, 1) RGBD_BCASE(2, 2, 3, 1) RGBD_BCASE(2, 2, 2, 1) RGBD_BCASE(2, 2, 1, 1)
    RGBD_BCASE(1, 4, 3, 2) RGBD_BCASE(1, 4, 2, 2) RGBD_BCASE(1, 4, 1, 2)
    RGBD_BCASE(1, 4, 3, 1) RGBD_BCASE(1, 4, 2, 1) RGBD_BCASE(1, 4, 1, 1)
    return RGBD_EINVAL;
}

// conv + fused trailing 1x1 (+ the next block's leading 1x1), blocked first layer: 128- and 64-pixel tiles
int launch_conv_fused_blk(const ConvArgs& a, int cls, hipStream_t s)
{
    if (a.w3) {
        if (cls >= 2) return launch_cfg<1, 4, 6, 2, 16, true, 2, true, 2, true>(a, pick_tw_log2(a.GW, a.GH, 128), s);
        if (cls == 1) return launch_cfg<1, 4, 6, 1, 16, true, 2, true, 2, true>(a, pick_tw_log2(a.GW, a.GH, 64), s);
        return RGBD_EINVAL;
    }
    if (cls >= 2) return launch_cfg<1, 4, 6, 2, 16, true, 3, false, 2, true>(a, pick_tw_log2(a.GW, a.GH, 128), s);
    if (cls == 1) return launch_cfg<1, 4, 6, 1, 16, true, 3, false, 2, true>(a, pick_tw_log2(a.GW, a.GH, 64), s);
    return RGBD_EINVAL;
}
