"""TEST INFRASTRUCTURE ONLY -- MS-SSIM from its published definition, in numpy fp64, as a cross-check of the product's
torch implementation (learning-based-rgb-d-image-compression_amd/metrics.py).

The reference calls `pytorch_msssim.ms_ssim(a, b, data_range=1)` (utils/metrics.py:13); that package
(pytorch-msssim==1.0.0, requirements.txt:48) is a third-party dependency that is neither vendored in /root/reference nor
installed in this image, so it cannot be run here: PARITY UNPINNED against the package itself.  What is restated is the
algorithm it documents: Wang, Simoncelli, Bovik, "Multiscale structural similarity for image quality assessment" (2003)
with the package's conventions -- 11-tap Gaussian window (sigma 1.5) applied separably WITHOUT padding, K1 = 0.01,
K2 = 0.03, five scales with weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333), 2x2 average pooling between scales (odd
sides zero-padded by one and divided by 4 like F.avg_pool2d with count_include_pad), contrast-structure terms clamped at
zero, per-channel product over scales, then the mean over channels and batch.
"""
import numpy as np

WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _window(size=11, sigma=1.5):
    c = np.arange(size, dtype=np.float64) - size // 2
    g = np.exp(-(c ** 2) / (2.0 * sigma ** 2))
    return g / g.sum()


def _blur(x, g):
    """valid separable correlation over the last two axes"""
    n = g.shape[0]
    H, W = x.shape[-2:]
    out = np.zeros(x.shape[:-2] + (H - n + 1, W), dtype=np.float64)
    for k in range(n):
        out += g[k] * x[..., k:k + H - n + 1, :]
    out2 = np.zeros(out.shape[:-1] + (W - n + 1,), dtype=np.float64)
    for k in range(n):
        out2 += g[k] * out[..., k:k + W - n + 1]
    return out2


def _pool2(x):
    H, W = x.shape[-2:]
    ph, pw = H % 2, W % 2
    if ph or pw:  # F.avg_pool2d(kernel 2, padding=(ph, pw)): zero pad both sides, windows of stride 2, divide by 4
        x = np.pad(x, [(0, 0)] * (x.ndim - 2) + [(ph, ph), (pw, pw)])
    H2, W2 = x.shape[-2] // 2, x.shape[-1] // 2
    x = x[..., :2 * H2, :2 * W2]
    return 0.25 * (x[..., 0::2, 0::2] + x[..., 1::2, 0::2] + x[..., 0::2, 1::2] + x[..., 1::2, 1::2])


def ms_ssim(a, b, data_range=1.0):
    x, y = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert x.shape == y.shape and x.ndim == 4
    if min(x.shape[-2:]) <= (11 - 1) * 2 ** 4:
        raise ValueError("image too small for 5-scale MS-SSIM (needs a side > 160)")
    g = _window()
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    terms = []
    for lvl in range(5):
        mu1, mu2 = _blur(x, g), _blur(y, g)
        s1 = _blur(x * x, g) - mu1 * mu1
        s2 = _blur(y * y, g) - mu2 * mu2
        s12 = _blur(x * y, g) - mu1 * mu2
        cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
        ssim_map = ((2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs_map
        if lvl < 4:
            terms.append(np.maximum(cs_map.mean(axis=(-2, -1)), 0.0))
            x, y = _pool2(x), _pool2(y)
        else:
            terms.append(np.maximum(ssim_map.mean(axis=(-2, -1)), 0.0))
    val = np.ones_like(terms[0])
    for t, w in zip(terms, WEIGHTS):
        val = val * t ** w
    return float(val.mean())
