"""sloika_amd -- MI355X (gfx950) native implementation of ONE hot path of nanoporetech/sloika:

    chunkify/normalise -> conv front end -> stacked GRU/LSTM -> softmax -> k-mer Viterbi decode

behind the reference's own layer / decode / worker API (see DESIGN.md and INTEGRATION.md).  The arithmetic runs in
hand-written HIP kernels (sloika_amd/csrc) loaded through a C ABI (include/sloika_amd.h); there is no CPU fallback.
"""
__version__ = "0.1.0"

import os as _os

# HIP maps streams onto 4 hardware queues unless told otherwise, and batches kept in flight on streams of their own (each with
# side streams for the directions of a birnn and one for copies) then serialise on them: baseline_raw_gru, 256 chunks, eight in
# flight ran 191 M samples/s on 4 queues and 417 M on 32.  The runtime reads the variable once, when it starts, so it is set
# here, at import, unless the user has chosen a value; device.want_hw_queues() warns when that came too late.
def _hip_already_running():
    """True when this process initialised the GPU runtime before importing sloika_amd (the value set below is then not read)."""
    import sys
    torch = sys.modules.get("torch")
    try:
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        return False


def _user_hw_queues():
    """The user's GPU_MAX_HW_QUEUES as an int (' 32', '+32' count), None when unset or not a number."""
    try:
        return int(_os.environ["GPU_MAX_HW_QUEUES"])
    except (KeyError, ValueError):
        return None


#: True when the user had set GPU_MAX_HW_QUEUES before this module was imported (recorded apart from the guess below)
HW_QUEUES_SET_BY_USER = _user_hw_queues() is not None
#: GPU_MAX_HW_QUEUES as the HIP runtime of this process saw (or will see) it: the user's value, else 32 if the runtime had not started
#: when this module was imported, else the runtime's default of 4.  Only torch's own initialisation is visible from here: a process that
#: started HIP by other means (a ctypes load, another framework, the rocprofv3 preload) is a case this guess gets wrong, which is why
#: device.want_hw_queues() words its warning for "unknown" rather than trusting 32 when torch is absent.
HW_QUEUES_IN_EFFECT = _user_hw_queues() if HW_QUEUES_SET_BY_USER else (4 if _hip_already_running() else 32)
#: False when the value above is a guess that could be wrong (no torch in the process to ask whether the runtime had started)
HW_QUEUES_KNOWN = HW_QUEUES_SET_BY_USER or ("torch" in __import__("sys").modules)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
