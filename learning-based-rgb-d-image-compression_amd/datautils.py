"""Padding / cropping of the reference harness (dataset/utils.py:41-100) on torch tensors (any device)."""
import torch.nn.functional as F


def pad0(x, p=2 ** 6, mode="reflect"):
    """Bottom/right padding to multiples of p (dataset/utils.py:58-67)."""
    H, W = x.size(2), x.size(3)
    pad_h = p * (H // p + 1) - H if H % p else 0
    pad_w = p * (W // p + 1) - W if W % p else 0
    return F.pad(x, (0, pad_w, 0, pad_h), mode=mode, value=0) if (pad_h or pad_w) else x


def pad1(x, p=2 ** 6, mode="reflect"):
    """Centred padding (dataset/utils.py:41-55)."""
    h, w = x.size(2), x.size(3)
    H, W = (h + p - 1) // p * p, (w + p - 1) // p * p
    left, top = (W - w) // 2, (H - h) // 2
    return F.pad(x, (left, W - w - left, top, H - h - top), mode=mode, value=0)


def crop0(x, size):
    return x[:, :, 0:size[0], 0:size[1]]


def crop1(x, size):
    H, W = x.size(2), x.size(3)
    h, w = size
    left, top = (W - w) // 2, (H - h) // 2
    return F.pad(x, (-left, -(W - w - left), -top, -(H - h - top)), mode="constant", value=0)


def pad(img, padding_mode, p=2 ** 6):
    """padding_mode is '<torch mode><0|1>', e.g. 'replicate0' (dataset/utils.py:92-100)."""
    if padding_mode.find("CenterCrop") != -1:
        raise NotImplementedError("CenterCrop((448, 576)) needs torchvision; the united tester uses replicate0")
    if padding_mode.find("0") != -1:
        return pad0(img, mode=padding_mode[:-1], p=p)
    return pad1(img, mode=padding_mode[:-1], p=p)


def crop(img, padding_mode, size):
    return crop0(img, size) if padding_mode.find("0") != -1 else crop1(img, size)
