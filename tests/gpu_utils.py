import pytest


def require_gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
