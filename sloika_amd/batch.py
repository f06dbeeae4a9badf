"""Chunk front end of the path (array maths of sloika/batch.py and sloika/tools/chunkify_raw.py), on the device.

    chunkify_signal(signal, chunk_len)             cut a read into [ml, chunk_len] chunks (chunkify_raw.py:172-176)
    normalise_chunks(chunks, 'per-chunk'|...)      median/MAD normalisation (chunkify_raw.py:178-185)
    chunks_to_network_input(chunks)                [ml, chunk_len] -> [chunk_len, ml, 1] (bin/train_network.py:304)
"""
import os
import threading

import numpy as np

from . import _lib, profiler

DEFAULT_NORMALISATION = 'per-read'
AVAILABLE_NORMALISATIONS = frozenset(['none', 'per-read', 'per-chunk'])


def chunkify_signal(signal, chunk_len):
    """First `ml*chunk_len` samples of `signal` as [ml, chunk_len] (chunkify_raw.py:172-176)."""
    assert len(signal) >= chunk_len
    ml = len(signal) // chunk_len
    return signal[: ml * chunk_len].reshape((ml, chunk_len))


def normalise_chunks(chunks, normalisation='per-chunk', out_layout='chunk', return_stats=False):
    """(x - median) / (1.4826 * MAD), per chunk or over the whole block ('per-read'), float32, bit-identical to
    the reference's numpy evaluation.

    chunks: [ml, chunk_len] numpy or device tensor.  out_layout 'chunk' -> [ml, chunk_len];
    'network' -> [chunk_len, ml, 1] (what Layer.run consumes) written directly by the kernel.
    """
    import torch
    from . import device as D
    assert normalisation in AVAILABLE_NORMALISATIONS
    cd = D.to_dev(chunks)
    if cd.dim() != 2:
        raise ValueError("chunks must be [nchunk, chunk_len]")
    ml, chunk_len = cd.shape
    if normalisation == 'none':
        res = cd if out_layout == 'chunk' else cd.t().contiguous()[:, :, None]
        return D.like_input(res, chunks)
    L = _lib.lib()
    if normalisation == 'per-read':
        # median / MAD of the whole ml*chunk_len block (chunkify_raw.py:182-183): one "chunk" of that length
        n_units, unit_len = 1, ml * chunk_len
    else:
        n_units, unit_len = ml, chunk_len
    med = D.scratch(n_units, torch.float32, cd.device)
    mad = D.scratch(n_units, torch.float32, cd.device)
    with profiler.region("normalise", 0.0, 8.0 * ml * chunk_len):
        if out_layout == 'chunk':
            out = D.scratch((ml, chunk_len), torch.float32, cd.device)
            rc = L.slk_med_mad_normalise_f32(cd.data_ptr(), n_units, unit_len, out.data_ptr(), unit_len, 1,
                                             med.data_ptr(), mad.data_ptr(), D.stream_ptr())
        else:
            if normalisation == 'per-read':
                tmp = torch.empty((ml, chunk_len), dtype=torch.float32, device=cd.device)
                rc = L.slk_med_mad_normalise_f32(cd.data_ptr(), 1, unit_len, tmp.data_ptr(), unit_len, 1,
                                                 med.data_ptr(), mad.data_ptr(), D.stream_ptr())
                out = tmp.t().contiguous()[:, :, None]
            else:
                out = D.scratch((chunk_len, ml, 1), torch.float32, cd.device)
                rc = L.slk_med_mad_normalise_f32(cd.data_ptr(), ml, chunk_len, out.data_ptr(), 1, ml,
                                                 med.data_ptr(), mad.data_ptr(), D.stream_ptr())
    _lib.check(rc, "normalise_chunks")
    res = D.like_input(out, chunks)
    if return_stats:
        return res, D.like_input(med, chunks), D.like_input(mad, chunks)
    return res


TRIM_OPEN_PORE_LOCAL_VAR_METHODS = frozenset(['mad', 'std'])


def trim_open_pore(signal, max_op_fraction=0.3, var_method='mad', window_size=100):
    """Locate raw read in signal by thresholding local variance (sloika/batch.py:194-220).

    The per-window MADs (the array maths of the reference: `maths.mad(sig_chunks, axis=1)`) are computed on the device
    by the normalisation kernel; the percentile threshold over those few hundred numbers and the slicing are host
    logic, as in the reference.  Returns a view of `signal`.

    :param signal: raw data containing a read (1D float32, numpy or device tensor)
    :param max_op_fraction: maximum expected fraction of signal that consists of open pore
    :param var_method: 'mad' (median absolute deviation, the default) or 'std' (standard deviation) of each window
    :param window_size: size of patches used to estimate local variance
    """
    import torch
    from . import device as D
    assert var_method in TRIM_OPEN_PORE_LOCAL_VAR_METHODS, "var_method not understood: {}".format(var_method)
    nwin = len(signal) // window_size
    sd = D.to_dev(signal)
    if sd.dim() != 1:
        raise ValueError("trim_open_pore expects a 1D signal")
    windows = sd[:nwin * window_size].reshape(nwin, window_size)
    if var_method == 'mad':
        _, _, spread = normalise_chunks(windows, 'per-chunk', return_stats=True)
    else:
        spread = torch.empty((nwin,), dtype=torch.float32, device=sd.device)
        _lib.check(_lib.lib().slk_window_std_f32(windows.contiguous().data_ptr(), nwin, window_size, spread.data_ptr(),
                                                 D.stream_ptr()), "window_std")
    spread = spread.cpu().numpy() if isinstance(spread, torch.Tensor) else np.asarray(spread)
    # windows livelier than the max_op_fraction quantile are read; keep everything from the first to the last of them
    lively = np.flatnonzero(spread > np.percentile(spread, 100 * max_op_fraction))
    first_win, last_win = int(lively[0]), int(lively[-1])
    return signal[first_win * window_size: (last_win + 1) * window_size]


def trim_open_pore_many(signals, max_op_fraction=0.3, var_method='mad', window_size=100):
    """trim_open_pore for a list of reads with ONE device call: the windows of all reads are concatenated, their MADs (or
    standard deviations) computed together, and the percentile threshold + slicing done per read on the host exactly as in
    trim_open_pore (sloika/batch.py:194-220).  Returns a list of views of the inputs."""
    import torch
    from . import device as D
    assert var_method in TRIM_OPEN_PORE_LOCAL_VAR_METHODS, "var_method not understood: {}".format(var_method)
    sigs = [np.asarray(s, dtype=np.float32) for s in signals]
    nwin = [len(s) // window_size for s in sigs]
    if min(nwin) < 1:
        raise ValueError("a read is shorter than one window of %d samples" % window_size)
    allw = np.concatenate([s[:n * window_size] for s, n in zip(sigs, nwin)]).reshape(-1, window_size)
    wd = D.to_dev(allw)
    if var_method == 'mad':
        _, _, spread = normalise_chunks(wd, 'per-chunk', return_stats=True)
    else:
        spread = torch.empty((wd.shape[0],), dtype=torch.float32, device=wd.device)
        _lib.check(_lib.lib().slk_window_std_f32(wd.data_ptr(), wd.shape[0], window_size, spread.data_ptr(), D.stream_ptr()),
                   "window_std")
    spread = spread.cpu().numpy()
    out, lo = [], 0
    for s, n in zip(sigs, nwin):
        sp = spread[lo:lo + n]
        lo += n
        lively = np.flatnonzero(sp > np.percentile(sp, 100 * max_op_fraction))
        out.append(s[int(lively[0]) * window_size: (int(lively[-1]) + 1) * window_size])
    return out


class _Staging(threading.local):
    """Pinned host staging buffer of upload_reads_windowed, one per host thread, kept between calls."""
    buf = None
    event = None


_staging = _Staging()


def upload_reads_windowed(signals, window_size=100):
    """All reads of a set in ONE upload (through pinned memory): read r occupies `dev[off[r] : off[r] + len[r]]`, every read padded
    with zeros to whole windows, so that `dev.view(-1, window_size)` is the window matrix of all reads.  -> (dev, off, lengths)."""
    import torch
    from . import device as D
    lens = [len(s) for s in signals]
    strides = [-(-n // window_size) * window_size for n in lens]
    off = np.concatenate([[0], np.cumsum(strides)]).astype(np.int64)
    total = int(off[-1])
    st = _staging
    if getattr(st, "buf", None) is None or st.buf.numel() < total:
        st.buf = torch.empty(max(total, 1 << 20), dtype=torch.float32).pin_memory()      # grow-only: pinning is the expensive part
        st.event = None
    if st.event is not None:
        st.event.synchronize()                           # the previous upload out of this buffer has left the host
    hv = st.buf.numpy()

    def pack(lo, hi):
        for r in range(lo, hi):
            hv[off[r]: off[r] + lens[r]] = signals[r]
            hv[off[r] + lens[r]: off[r + 1]] = 0.0

    n = len(signals)
    if total >= (1 << 24) and n >= 16:
        # a gigabyte of samples is a tenth of a second of memcpy on one core; numpy's copies release the interpreter lock
        import concurrent.futures
        nthr = min(8, os.cpu_count() or 1)
        cuts = np.searchsorted(off, np.linspace(0, total, nthr + 1)[1:-1]).tolist()
        edges = [0] + [min(max(c, 0), n) for c in cuts] + [n]
        with concurrent.futures.ThreadPoolExecutor(nthr) as ex:
            list(ex.map(lambda a: pack(*a), [(edges[k], edges[k + 1]) for k in range(nthr) if edges[k + 1] > edges[k]]))
    else:
        pack(0, n)
    dev = st.buf[:total].to(D.device(), non_blocking=True)
    st.event = torch.cuda.Event()
    st.event.record()
    return dev, off, lens


def open_pore_bounds_many(dev, off, lens, max_op_fraction=0.3, var_method='mad', window_size=100):
    """trim_open_pore (sloika/batch.py:194-220) for reads resident on the device as upload_reads_windowed leaves them: the
    spreads of ALL windows in one launch, the percentile threshold per read on the host (a few hundred numbers each).
    -> list of (first sample, one past the last sample) relative to each read's start; None for a read the reference's function
    would fail on (shorter than one window, or no window livelier than the threshold)."""
    import torch
    from . import device as D
    assert var_method in TRIM_OPEN_PORE_LOCAL_VAR_METHODS, "var_method not understood: {}".format(var_method)
    nwin = np.asarray([n // window_size for n in lens], dtype=np.int64)
    wd = dev.view(-1, window_size)
    if var_method == 'mad':
        _, _, spread = normalise_chunks(wd, 'per-chunk', return_stats=True)
    else:
        spread = torch.empty((wd.shape[0],), dtype=torch.float32, device=wd.device)
        _lib.check(_lib.lib().slk_window_std_f32(wd.data_ptr(), wd.shape[0], window_size, spread.data_ptr(), D.stream_ptr()),
                   "window_std")
    spread = spread.cpu().numpy()
    w0 = np.asarray(off[:-1] if len(off) == len(lens) + 1 else off, dtype=np.int64) // window_size
    out = [None] * len(lens)
    if max_op_fraction == 0 and len(lens):
        # np.percentile(., 0) is the minimum: all reads at once (whole windows only, as the reference's reshape leaves them)
        live = np.flatnonzero(nwin > 0)
        if len(live):
            cnt = nwin[live]
            seg0 = np.concatenate([[0], np.cumsum(cnt)[:-1]])
            pos = np.arange(int(cnt.sum()), dtype=np.int64) - np.repeat(seg0, cnt)          # window index inside its read
            vals = spread[np.repeat(w0[live], cnt) + pos]
            lively = vals > np.repeat(np.minimum.reduceat(vals, seg0), cnt)
            first = np.minimum.reduceat(np.where(lively, pos, np.iinfo(np.int64).max), seg0)
            last = np.maximum.reduceat(np.where(lively, pos, -1), seg0)
            for k, r in enumerate(live):
                if last[k] >= 0:
                    out[r] = (int(first[k]) * window_size, (int(last[k]) + 1) * window_size)
        return out
    for r, n in enumerate(nwin):
        if n < 1:
            continue
        sp = spread[w0[r]: w0[r] + n]                    # whole windows only, as the reference's reshape leaves them
        lively = np.flatnonzero(sp > np.percentile(sp, 100 * max_op_fraction))
        if len(lively):
            out[r] = (int(lively[0]) * window_size, (int(lively[-1]) + 1) * window_size)
    return out


def reads_nonfinite(dev, off, lens):
    """Which reads of an uploaded read set (upload_reads_windowed) hold a NaN or an infinity: bool array [nread]."""
    import torch
    from . import device as D
    n = len(lens)
    if n == 0:
        return np.zeros(0, dtype=bool)
    start = torch.as_tensor(np.ascontiguousarray(off[:n], dtype=np.int64)).to(dev.device)
    ln = torch.as_tensor(np.ascontiguousarray(lens, dtype=np.int32)).to(dev.device)
    flags = torch.zeros((n,), dtype=torch.int32, device=dev.device)
    for lo in range(0, n, 65535):
        hi = min(n, lo + 65535)
        _lib.check(_lib.lib().slk_reads_nonfinite_f32(dev.data_ptr(), start[lo:].data_ptr(), ln[lo:].data_ptr(), hi - lo,
                                                      int(max(lens[lo:hi])), flags[lo:].data_ptr(), D.stream_ptr()), "reads_nonfinite")
    return flags.cpu().numpy() != 0


def normalise_reads_ragged(padded, lengths):
    """Median/MAD normalisation of whole reads of different lengths over their OWN lengths (sloika/basecall.py:117-118) in one
    launch: `padded` is a [B, Lmax] float32 device tensor (read b in its first lengths[b] samples), `lengths` an int32 device
    tensor [B].  -> [Lmax, B, 1] network layout, zero behind every read's end."""
    import torch
    from . import device as D
    B, lmax = padded.shape
    out = D.scratch((lmax, B, 1), torch.float32, padded.device).zero_()
    with profiler.region("normalise", 0.0, 8.0 * B * lmax):
        rc = _lib.lib().slk_med_mad_normalise_ragged_f32(padded.data_ptr(), B, padded.stride(0), lengths.data_ptr(), out.data_ptr(),
                                                         1, B, None, None, D.stream_ptr())
    _lib.check(rc, "normalise_reads_ragged")
    return out


def chunks_to_network_input(chunks):
    """[ml, chunk_len] -> [chunk_len, ml, 1] (the transpose of bin/train_network.py:304)."""
    import torch
    if isinstance(chunks, torch.Tensor):
        return chunks.t().contiguous()[:, :, None]
    return np.ascontiguousarray(np.asarray(chunks).T)[:, :, None]


#: process-global set by init_chunk_identity_worker / init_chunk_remap_worker (sloika/batch.py:127-140)
kmer_to_state = None
kmer_alphabet = None
calc_post = None


def init_chunk_identity_worker(kmer_len, alphabet):
    """sloika/batch.py:127-129: the k-mer -> state dictionary of this process (and the alphabet it was built from, which the
    device-side label kernels take instead of a dictionary)."""
    global kmer_to_state, kmer_alphabet
    from . import bio
    kmer_to_state = bio.kmer_mapping(kmer_len, alphabet=alphabet)
    kmer_alphabet = alphabet if isinstance(alphabet, bytes) else alphabet.encode('ascii')


def init_chunk_remap_worker(model, kmer_len, alphabet):
    """sloika/batch.py:132-140: as above, plus the compiled model in the process-global `calc_post`.  `model` is a model file
    name or a Layer."""
    global calc_post
    from . import helpers, layers
    init_chunk_identity_worker(kmer_len, alphabet)
    net = model if isinstance(model, layers.Layer) else helpers.load_model(model)
    calc_post = net.compile()
