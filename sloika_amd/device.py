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


class Arena(object):
    """Device buffers a caller keeps across calls, handed out in call order: the k-th request of a pass gets the k-th buffer
    (grown when the request is bigger than anything it held before), so a pass whose shapes repeat allocates nothing after the
    first.  This is the reference's `borrow=True` contract (layers.py:34-36: th.function(In(borrow=True), Out(borrow=True)) -- the
    returned buffer may be overwritten by the next call, the caller consumes or copies it first) with one relaxation: requests
    marked `result` rotate over `generations` sets, so what a call returns stays intact until `generations` further calls have
    been issued (the bench copies paths to the host on a copy stream while the next call already runs).

    One arena belongs to one stream of work (pipeline.Basecaller owns one): passes on it are issued in order on one stream, which
    is what makes handing the same memory to the next pass safe."""

    def __init__(self, generations=2):
        self.bufs, self.cursor = [], 0
        self.res, self.rcursor, self.gen = [[] for _ in range(max(1, generations))], 0, 0
        self.grown = 0                    # allocations so far (tests: a warm pass adds none)

    def begin(self):
        self.cursor = self.rcursor = 0
        self.gen = (self.gen + 1) % len(self.res)

    def take(self, shape, dtype, dev, result=False):
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        n = 1
        for v in shape:
            n *= v
        esz = _ESIZE.get(dtype)
        if esz is None:
            esz = _ESIZE[dtype] = torch.empty((), dtype=dtype).element_size()
        nbytes = n * esz
        pool = self.res[self.gen] if result else self.bufs
        k = self.rcursor if result else self.cursor
        if result:
            self.rcursor += 1
        else:
            self.cursor += 1
        if k == len(pool):
            pool.append(None)
        buf = pool[k]
        if buf is None or buf.numel() < nbytes or buf.device != dev:
            pool[k] = None
            buf = pool[k] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
            self.grown += 1
        return buf[:nbytes].view(dtype).view(shape)

    def __enter__(self):
        self._outer = getattr(_ARENA, "cur", None)
        _ARENA.cur = self
        self.begin()
        return self

    def __exit__(self, *exc):
        _ARENA.cur = self._outer
        return False


import threading  # noqa: E402
_ARENA = threading.local()
_ESIZE = {}                                  # dtype -> bytes per element


def scratch(shape, dtype=torch.float32, dev=None, result=False):
    """An uninitialised device tensor for the pass in progress: out of the active Arena (`with arena:`) when there is one, a fresh
    torch.empty otherwise.  `result`: the tensor is handed back to the caller of the pass (see Arena)."""
    dev = device() if dev is None else dev
    cur = getattr(_ARENA, "cur", None)
    if cur is None:
        return torch.empty(shape, dtype=dtype, device=dev)
    return cur.take(shape, dtype, dev, result)


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
    if not sloika_amd.HW_QUEUES_KNOWN and have >= min(n, 32):
        # nobody in this process could tell whether HIP had started before sloika_amd set the variable (no torch at import: a ctypes
        # load, another framework or the rocprofv3 preload may have started it with the default of 4)
        warnings.warn("sloika_amd: GPU_MAX_HW_QUEUES in effect is unknown (may be the runtime's default of 4); %d streams are meant to "
                      "run side by side -- set GPU_MAX_HW_QUEUES=32 in the environment to be sure" % n)
    if have < min(n, 32):
        warnings.warn("sloika_amd: GPU_MAX_HW_QUEUES=%d but %d streams are meant to run side by side; batches in flight will "
                      "serialise on the hardware queues (set GPU_MAX_HW_QUEUES=32 before the process touches the GPU)" % (have, n))
    return have
