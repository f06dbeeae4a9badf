"""decode API of the reference (sloika/decode.py) on the HIP kernels of csrc/decode.hip.

    argmax(post, zero_is_blank=True)                       decode.py:5-18
    prepare_post(post, min_prob=1e-5, drop_bad=False)      decode.py:21-36
    viterbi(post, klen, skip_pen=0.0, log=False, nbase=4)  decode.py:39-93      -> (score, [states])
    viterbi_batch(post[T,B,S], ...)                        batched extension    -> (scores[B], paths[B,T], lens[B])

Inputs may be numpy arrays or device tensors; float32 arithmetic (the network's dtype).  float64 input is
converted to float32 first -- unlike numpy, which would then decode in float64.
"""
import numpy as np

from . import _lib, profiler
from . import variables as sv


def _dev(x):
    from . import device as D
    return D.to_dev(x)


def argmax(post, zero_is_blank=True):
    """Argmax decoding of simple transducer (decode.py:5-18): 1D array of called states."""
    import torch
    from . import device as D
    pd = _dev(post)
    assert pd.dim() == 2
    T, S = pd.shape
    path = torch.empty((1, T), dtype=torch.int32, device=pd.device)
    n = torch.empty(1, dtype=torch.int32, device=pd.device)
    _lib.check(_lib.lib().slk_argmax_decode_f32(pd.data_ptr(), T, 1, S, int(bool(zero_is_blank)), path.data_ptr(),
                                                n.data_ptr(), D.stream_ptr()), "decode.argmax")
    return path[0, : int(n.item())].cpu().numpy().astype(np.int64)


def prepare_post(post, min_prob=1e-5, drop_bad=False):
    """Sanitised posterior matrix for decoding (decode.py:21-36): [T,1,S] -> [T,S]."""
    import torch
    from . import device as D
    if drop_bad:
        raise NotImplementedError("drop_bad=True belongs to the non-transducer decoder (sloika/olddecode.py), "
                                  "which is outside the accelerated path")
    pd = _dev(post)
    if pd.dim() != 3 or pd.shape[1] != 1:
        raise ValueError("prepare_post expects a [time, 1, state] posterior (np.squeeze(axis=1), decode.py:30)")
    pd = pd[:, 0, :].contiguous()
    out = torch.empty_like(pd)
    _lib.check(_lib.lib().slk_prepare_post_f32(pd.data_ptr(), out.data_ptr(), pd.numel(), float(min_prob),
                                               D.stream_ptr()), "decode.prepare_post")
    return D.like_input(out, post)


class ViterbiWorkspace(object):
    """Reusable device buffers for viterbi_batch (traceback bytes dominate: T*B*nkmer)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        import torch
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = None
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.buf


def viterbi_batch(post, klen, skip_pen=0.0, log=False, nbase=4, min_prob=None, workspace=None):
    """Batched decode.viterbi over the batch axis of a [T, B, nstate] posterior.

    min_prob: if given, decode.prepare_post's transform is applied first (fused), i.e. `post` is the raw
    network output as in basecall.decode_post (sloika/basecall.py:26-51).
    Returns device tensors (scores float32 [B], paths int32 [B, T] left aligned and -1 padded, lens int32 [B]).
    """
    import torch
    from . import device as D
    pd = _dev(post)
    if pd.dim() != 3:
        raise ValueError("viterbi_batch expects [time, batch, state]")
    T, B, S = pd.shape
    if klen < 3:
        raise ValueError("Kmer not long enough to apply Viterbi with skips")          # decode.py:50
    if sv.nstate(klen, transducer=True, nbase=nbase) != S:
        raise ValueError("posterior has %d states, klen=%d nbase=%d needs %d" % (S, klen, nbase, sv.nstate(klen, nbase=nbase)))
    L = _lib.lib()
    mode = _lib.POST_LOG if log else (_lib.POST_RAW if min_prob is not None else _lib.POST_PLAIN)
    nbytes = L.slk_viterbi_kmer_workspace_bytes(T, B, nbase, klen)
    if nbytes == 0:
        raise ValueError("unsupported klen/nbase for the Viterbi kernel")
    ws = (workspace or ViterbiWorkspace()).get(nbytes, pd.device)
    scores = D.scratch(B, torch.float32, pd.device, result=True)
    paths = D.scratch((B, T), torch.int32, pd.device, result=True)
    lens = D.scratch(B, torch.int32, pd.device, result=True)
    nk = nbase ** klen
    with profiler.region("viterbi", 0.0, float(T) * B * (4.0 * S + 2.0 * nk)):
        rc = L.slk_viterbi_kmer_f32(pd.data_ptr(), T, B, nbase, klen, float(skip_pen), mode,
                                    float(min_prob if min_prob is not None else 0.0), ws.data_ptr(), nbytes,
                                    scores.data_ptr(), paths.data_ptr(), lens.data_ptr(), D.stream_ptr())
    _lib.check(rc, "decode.viterbi")
    return scores, paths, lens


def viterbi_logits_batch(logits, stats, klen, T, B, ld=None, skip_pen=0.0, nbase=4, min_prob=1e-5, workspace=None,
                         lengths=None):
    """basecall.decode_post over the batch axis, fed with the Softmax layer's LOGITS (rows (t,b) of nstate floats, `ld`
    floats apart) and their row statistics (layers.Softmax.logits_and_stats): softmax, prepare_post, log and Viterbi in one pass over the logits.
    Bit-identical to viterbi_batch(softmax(logits), min_prob=min_prob).  `lengths`: optional int32 device tensor [B] for a
    ragged batch (chunk b decoded over its first lengths[b] steps only)."""
    import torch
    from . import device as D
    if klen < 3:
        raise ValueError("Kmer not long enough to apply Viterbi with skips")
    S = sv.nstate(klen, transducer=True, nbase=nbase)
    ld = S if ld is None else ld
    if ld < S or logits.numel() < T * B * ld:
        raise ValueError("logits buffer too small for T=%d B=%d ld=%d" % (T, B, ld))
    L = _lib.lib()
    nbytes = L.slk_viterbi_kmer_workspace_bytes(T, B, nbase, klen)
    if nbytes == 0:
        raise ValueError("unsupported klen/nbase for the Viterbi kernel")
    ws = (workspace or ViterbiWorkspace()).get(nbytes, logits.device)
    scores = D.scratch(B, torch.float32, logits.device, result=True)
    paths = D.scratch((B, T), torch.int32, logits.device, result=True)
    lens = D.scratch(B, torch.int32, logits.device, result=True)
    nk = nbase ** klen
    with profiler.region("viterbi", 0.0, float(T) * B * (4.0 * S + 2.0 * nk)):
        if lengths is None:
            rc = L.slk_viterbi_kmer_logits_f32(logits.data_ptr(), ld, stats.data_ptr(), T, B, nbase, klen, float(skip_pen),
                                               float(min_prob), ws.data_ptr(), nbytes, scores.data_ptr(),
                                               paths.data_ptr(), lens.data_ptr(), D.stream_ptr())
        else:
            if lengths.dtype != torch.int32 or lengths.numel() != B or not lengths.is_cuda:
                raise ValueError("lengths must be an int32 device tensor with one entry per chunk")
            rc = L.slk_viterbi_kmer_logits_ragged_f32(logits.data_ptr(), ld, stats.data_ptr(), T, B, nbase, klen,
                                                      float(skip_pen), float(min_prob), lengths.data_ptr(), ws.data_ptr(),
                                                      nbytes, scores.data_ptr(), paths.data_ptr(), lens.data_ptr(),
                                                      D.stream_ptr())
    _lib.check(rc, "decode.viterbi_logits")
    return scores, paths, lens


def viterbi_fused_batch(x, pack, klen, skip_pen=0.0, nbase=4, min_prob=1e-5, workspace=None, lengths=None, lp_dump=None, plan=0):
    """basecall.decode_post(softmax layer(x)) over the batch axis from the Softmax layer's INPUT x [T, B, insize] and its
    packed weights (layers.Softmax.viterbi_pack): projection, softmax (layers.py:309-314), prepare_post (decode.py:21-36),
    log and the Viterbi forward pass (decode.py:39-82) in one kernel (csrc/softmax_viterbi.hip), then the backtrace
    (decode.py:84-91).  The logits are never written.  `lp_dump`: optional float32 device tensor [T, B, nstate] that
    receives the log-posteriors the dynamic programme consumed.  `plan`: chunks per workgroup (2, 4, or 0 = by batch size; the
    results do not depend on it).  Same outputs as viterbi_logits_batch."""
    import torch
    from . import device as D
    if klen < 3:
        raise ValueError("Kmer not long enough to apply Viterbi with skips")
    if x.dim() != 3 or x.dtype != torch.float32 or not x.is_cuda or x.stride(2) != 1 or x.stride(0) != x.shape[1] * x.stride(1):
        raise ValueError("x must be a float32 device tensor [time, batch, features] with uniformly spaced rows")
    T, B, K = x.shape
    S = sv.nstate(klen, transducer=True, nbase=nbase)
    L = _lib.lib()
    nbytes = L.slk_softmax_viterbi_workspace_bytes(T, B, nbase, klen)      # one traceback byte per four k-mers: 256 B per (step, chunk)
    if nbytes == 0:
        raise ValueError("unsupported klen/nbase for the fused softmax + Viterbi kernel")
    ws = (workspace or ViterbiWorkspace()).get(nbytes, x.device)
    scores = D.scratch(B, torch.float32, x.device, result=True)
    paths = D.scratch((B, T), torch.int32, x.device, result=True)
    lens = D.scratch(B, torch.int32, x.device, result=True)
    if lengths is not None and (lengths.dtype != torch.int32 or lengths.numel() != B or not lengths.is_cuda):
        raise ValueError("lengths must be an int32 device tensor with one entry per chunk")
    if lp_dump is not None and (lp_dump.dtype != torch.float32 or lp_dump.numel() != T * B * S or not lp_dump.is_contiguous()):
        raise ValueError("lp_dump must be a contiguous float32 device tensor [T, B, nstate]")
    rows = float(T) * B
    # bytes: the rows of x in, one traceback byte per four k-mers out (csrc/softmax_viterbi.hip, D8) and at most the same again for the walk
    with profiler.region("softmax_viterbi", 2.0 * rows * K * S, rows * (4.0 * K + 0.5 * (nbase ** klen)),
                         f16x3_flops=2.0 * rows * K * S):
        rc = L.slk_softmax_viterbi_f32(x.data_ptr(), x.stride(1), pack.data_ptr(), K, T, B, nbase, klen, float(skip_pen),
                                       float(min_prob), lengths.data_ptr() if lengths is not None else None, int(plan), ws.data_ptr(),
                                       nbytes, scores.data_ptr(), paths.data_ptr(), lens.data_ptr(),
                                       lp_dump.data_ptr() if lp_dump is not None else None, D.stream_ptr())
    _lib.check(rc, "decode.viterbi_fused")
    return scores, paths, lens


def viterbi(post, klen, skip_pen=0.0, log=False, nbase=4):
    """Viterbi decoding of a kmer transducer (decode.py:39-93).

    :param post: A 2d array [time, nstate]
    :param klen: Length of kmer
    :param log: post array is in log space

    :returns: (score, list of k-mer states)
    """
    pd = _dev(post)
    if pd.dim() != 2:
        raise ValueError("viterbi expects a [time, state] posterior")
    scores, paths, lens = viterbi_batch(pd[:, None, :], klen, skip_pen=skip_pen, log=log, nbase=nbase)
    n = int(lens[0].item())
    return np.float32(scores[0].item()), [int(v) for v in paths[0, :n].cpu().numpy()]
