"""Device-memory plumbing: torch is used ONLY for HBM allocation, streams and host<->device copies."""
import numpy as np
import torch

from . import _lib


def device():
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    """hipStream_t of torch's current stream, as an integer for ctypes."""
    return torch.cuda.current_stream().cuda_stream


def to_dev(x, dtype=torch.float32):
    """numpy / torch -> contiguous device tensor of `dtype` (no copy if it already is one)."""
    dev = device()
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=dtype).contiguous()
    a = np.ascontiguousarray(x)
    return torch.from_numpy(a).to(device=dev, dtype=dtype)


def empty(shape, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=device())


def ptr(t):
    return None if t is None else t.data_ptr()


def like_input(result, template):
    """Return `result` as numpy when the caller passed numpy (the reference's compiled functions take and
    return ndarrays, layers.py:34-36), otherwise leave it on the device."""
    if isinstance(template, torch.Tensor):
        return result
    return result.cpu().numpy()


def want_hw_queues(n):
    """Called by users of several streams (pipeline.Basecaller(in_flight > 2)): warn when the HIP runtime was started with fewer
    hardware queues than streams that are meant to run side by side (GPU_MAX_HW_QUEUES, default 4, is read once at start-up:
    sloika_amd sets 32 at import unless the environment says otherwise)."""
    import warnings
    import sloika_amd
    have = sloika_amd.HW_QUEUES_IN_EFFECT            # (not os.environ: the import sets the variable even when it comes too late)
    if have < min(n, 32):
        warnings.warn("sloika_amd: GPU_MAX_HW_QUEUES=%d but %d streams are meant to run side by side; batches in flight will "
                      "serialise on the hardware queues (set GPU_MAX_HW_QUEUES=32 before the process touches the GPU)" % (have, n))
    return have
