"""Per-stage device timing with HIP events on the launch stream (torch's current stream is the stream every
C-ABI call is enqueued on).  Disabled by default: `region()` is then a no-op context manager.

Each region carries the ALGORITHMIC flops and bytes of the launches inside it (see DESIGN.md "Kernels"), so that
bench.py can turn the measured duration into a roofline fraction.
"""
import contextlib

_active = None


class Recorder(object):
    def __init__(self):
        self.records = []          # (name, start_event, end_event, flops, bytes, flops as 3-term fp16 split, flops as 2-MFMA split)

    def summary(self):
        """name -> dict(calls, ms_total, ms_avg, flops, f16x3_flops, f16x2_flops, bytes) ; call after torch.cuda.synchronize()."""
        out = {}
        for name, e0, e1, flops, nbytes, f16, f16x2 in self.records:
            d = out.setdefault(name, {"calls": 0, "ms_total": 0.0, "flops": 0.0, "bytes": 0.0, "f16x3_flops": 0.0, "f16x2_flops": 0.0})
            d["calls"] += 1
            d["ms_total"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
            d["f16x3_flops"] += f16
            d["f16x2_flops"] += f16x2
        for d in out.values():
            d["ms_avg"] = d["ms_total"] / d["calls"]
        return out


def start():
    global _active
    _active = Recorder()
    return _active


def stop():
    global _active
    rec, _active = _active, None
    return rec


class _Region(object):
    def __init__(self):
        self.cancelled = False

    def cancel(self):
        """Drop this region (e.g. the call inside turned out to be a no-op that fell back to another path)."""
        self.cancelled = True


@contextlib.contextmanager
def region(name, flops=0.0, nbytes=0.0, f16x3_flops=0.0, f16x2_flops=0.0):
    """`f16x3_flops`: the part of `flops` that the kernel evaluates as three fp16 MFMAs per product (fp32-grade split
    arithmetic on the fp16 pipe) -- bench.py prices that part against the fp16 matrix peak, the rest against the fp32 one.
    `f16x2_flops`: the part evaluated with TWO fp16 MFMAs per product (the recurrent products of the Gru kernels whose spare
    MFMA columns carry the lo half of the state, csrc/bar16_common.h) -- not included in `f16x3_flops`."""
    if _active is None:
        yield None
        return
    import torch
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    reg = _Region()
    e0.record()
    try:
        yield reg
    finally:
        e1.record()
        if not reg.cancelled:
            _active.records.append((name, e0, e1, float(flops), float(nbytes), float(f16x3_flops), float(f16x2_flops)))
