"""Quality metrics of the reference harness (utils/metrics.py:8-30).

PSNR follows the reference exactly (MSE of the clamped tensors).  MS-SSIM restates the published definition that
pytorch-msssim 1.0.0 (requirements.txt:48) implements (11-tap Gaussian, sigma 1.5, five scales, default weights); that
package is not installable in the build image, so the number is not pinned against the package itself -- it is checked
against an independent numpy fp64 statement of the definition (oracle/msssim_ref.py, tests/test_host_logic.py) and the
harness log says so.
"""
import numpy as np
import torch
import torch.nn.functional as F

_MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def psnr(a: torch.Tensor, b: torch.Tensor, max_val: float = 1.0) -> float:
    mse = torch.mean((a.clamp(0, 1) - b.clamp(0, 1)) ** 2).item()
    return float(20 * np.log10(max_val) - 10 * np.log10(mse))


def _gauss(size=11, sigma=1.5, device=None):
    c = torch.arange(size, dtype=torch.float32, device=device) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _filter(x, g):
    C = x.shape[1]
    x = F.conv2d(x, g.view(1, 1, -1, 1).repeat(C, 1, 1, 1), groups=C)
    return F.conv2d(x, g.view(1, 1, 1, -1).repeat(C, 1, 1, 1), groups=C)


def _ssim_cs(x, y, g, data_range):
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu1, mu2 = _filter(x, g), _filter(y, g)
    s1 = _filter(x * x, g) - mu1 * mu1
    s2 = _filter(y * y, g) - mu2 * mu2
    s12 = _filter(x * y, g) - mu1 * mu2
    cs = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim = ((2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs
    return ssim.flatten(2).mean(-1), cs.flatten(2).mean(-1)


def ms_ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> float:
    return float(ms_ssim_tensor(x, y, data_range).item())


def ms_ssim_gpu(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0, clamp01: bool = False) -> torch.Tensor:
    """MS-SSIM per image of a batch, [N] on the device, through the HIP library (csrc/metrics.hip: one call does the five
    scales of all N * C planes; the torch form below is ~60 small launches per image enqueued under the GIL, which was
    most of the pipelined harness's wall time).  Same definition; agrees with the torch form / the fp64 statement to
    < 2e-5 (tests/test_gpu_harness.py)."""
    import ctypes

    from ._lib import check, lib

    if min(x.shape[-2:]) <= (11 - 1) * 2 ** 4:
        raise ValueError("image too small for 5-scale MS-SSIM (needs a side > 160)")
    x, y = x.float().contiguous(), y.float().contiguous()
    N, C, H, W = x.shape
    P = N * C
    L = lib()
    nbytes = int(L.rgbd_msssim_workspace_bytes(P, H, W))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    out = torch.empty((N, C, 5, 2), dtype=torch.float32, device=x.device)
    taps = _gauss().numpy()  # (CPU, the same eleven floats the torch form filters with)
    check(L.rgbd_msssim_stats(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), P, H, W,
                              taps.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), float(data_range), 1 if clamp01 else 0,
                              ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()), nbytes,
                              ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "msssim_stats")
    w = torch.tensor(_MS_WEIGHTS, device=x.device)
    vals = torch.cat([torch.relu(out[:, :, :4, 1]), torch.relu(out[:, :, 4:, 0])], dim=2)  # cs of scales 0-3, ssim of scale 4
    return torch.prod(vals ** w, dim=2).mean(dim=1)


def ms_ssim_tensor(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """The value as a 0-d tensor on x's device (no host synchronisation: the pipelined harness fetches all of an image's
    metrics with one copy).  CUDA tensors go through the HIP kernel (ms_ssim_gpu), CPU tensors through the torch form."""
    if x.is_cuda:
        return ms_ssim_gpu(x, y, data_range).mean() if x.shape[0] == 1 else _ms_ssim_torch(x, y, data_range)
    return _ms_ssim_torch(x, y, data_range)


def _ms_ssim_torch(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    if min(x.shape[-2:]) <= (11 - 1) * 2 ** 4:
        raise ValueError("image too small for 5-scale MS-SSIM (needs a side > 160)")
    g = _gauss(device=x.device)
    w = torch.tensor(_MS_WEIGHTS, device=x.device)
    mcs = []
    for i in range(5):
        s, cs = _ssim_cs(x, y, g, data_range)
        if i < 4:
            mcs.append(torch.relu(cs))
            pad = [d % 2 for d in x.shape[2:]]
            x = F.avg_pool2d(x, 2, padding=pad)
            y = F.avg_pool2d(y, 2, padding=pad)
    vals = torch.stack(mcs + [torch.relu(s)], dim=0)
    return torch.prod(vals ** w.view(-1, 1, 1), dim=0).mean()


def compute_metrics(a, b, max_val: float = 1.0):
    a, b = a.clamp(0, 1), b.clamp(0, 1)
    p = psnr(a, b, max_val)
    try:
        m = ms_ssim(a.float(), b.float(), data_range=max_val)
    except ValueError:
        m = float("nan")
    return p, m


def metrics_batch(a, b, max_val: float = 1.0) -> torch.Tensor:
    """[N, 2] = per image [mse, ms_ssim (nan when the images are too small)] of the clamped tensors on the device: two tensor
    ops for the mse and ONE library call for MS-SSIM (the pipelined harness: a group of images per call)."""
    a, b = a.clamp(0, 1), b.clamp(0, 1)
    mse = torch.stack([torch.mean((a[i:i + 1] - b[i:i + 1]) ** 2) for i in range(a.shape[0])])  # (per image, as psnr() does)
    try:
        m = ms_ssim_gpu(a, b, max_val) if a.is_cuda else torch.stack([_ms_ssim_torch(a[i:i + 1].float(), b[i:i + 1].float(), max_val)
                                                                       for i in range(a.shape[0])])
    except ValueError:
        m = torch.full_like(mse, float("nan"))
    return torch.stack([mse, m], dim=1)


def metrics_tensor(a, b, max_val: float = 1.0) -> torch.Tensor:
    """[mse, ms_ssim (nan when the image is too small)] of the clamped tensors as one device tensor; `finish_metrics` turns
    the fetched pair into compute_metrics()'s (psnr, ms_ssim) with the same host arithmetic."""
    a, b = a.clamp(0, 1), b.clamp(0, 1)
    mse = torch.mean((a - b) ** 2)
    try:
        m = ms_ssim_tensor(a.float(), b.float(), data_range=max_val)
    except ValueError:
        m = torch.full_like(mse, float("nan"))
    return torch.stack([mse, m])


def finish_metrics(mse: float, m: float, max_val: float = 1.0):
    return float(20 * np.log10(max_val) - 10 * np.log10(mse)), float(m)


class AverageMeter:
    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
