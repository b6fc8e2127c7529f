// measured by tools/tune_splitk.py on MI355X (c3: 4 x 512x640, RGB / depth pairs as one launch): split-K factors that beat
// the rule of conv_splitk_for by > 6 % (kernel + reducer, best tile on both sides).  A function of the layer and the
// per-image output grid only.
// out_px_per_image, cin_pad, cout_pad, ntaps, nphase,   S
{80, 192, 192, 25, 1,   6},
{320, 192, 192, 25, 1,   2},
{1280, 48, 32, 25, 1,   3},
{1280, 96, 64, 25, 1,   3},
{1280, 128, 32, 25, 1,   3},
{1280, 224, 48, 9, 1,   4},
{1280, 240, 48, 9, 1,   3},
{1280, 240, 96, 9, 1,   3},
{1280, 256, 96, 9, 1,   3},
{1280, 384, 320, 25, 1,   5},
