"""Helpers shared by the `-m gpu` parity tests (all of them call the product through its C ABI)."""
import numpy as np
import pytest


def need_gpu():
    import torch
    from sloika_amd import _lib
    if not torch.cuda.is_available() or _lib.lib().slk_device_count() < 1:
        pytest.fail("gpu-marked test needs an AMD GPU and libsloika_amd.so (no CPU fallback exists)")
    return torch


def dev(a, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
