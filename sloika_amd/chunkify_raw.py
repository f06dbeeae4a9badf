"""Remapping a read to its reference and cutting it into labelled training chunks (sloika/tools/chunkify_raw.py), with the
array work on the device.

    raw_remap(ref, signal, min_prob, kmer_len, prior, slip)            chunkify_raw.py:260-296
    raw_remap_many(refs, signals, ...)                                 the same for many reads, one remap launch
    raw_chunkify(signal, mapping_table, chunk_len, kmer_len, ...)      chunkify_raw.py:164-210
    raw_chunkify_many(signals, mapping_tables, ...)                    the same for many reads, one label launch
    raw_chunk_remap_worker(fn, ...)                                    chunkify_raw.py:299-337
    mapping-table helpers                                              chunkify_raw.py:18-161

What runs where: the mapping table itself is a small numpy record array in the reference's API (one row per network
output step) and stays one; trimming / registering it is host bookkeeping on a few thousand rows.  Everything that
touches samples or posteriors -- median/MAD normalisation, the network, prepare_post, the remap DP with its slip scan,
k-mer -> state lookups and the per-chunk label generation -- runs through the C ABI (csrc/frontend.hip, transducer.hip,
chunk_labels.hip).  There is no CPU version of those steps here.
"""
import sys

import numpy as np

from . import _lib, batch, bio, decode, transducer, util
from .batch import AVAILABLE_NORMALISATIONS, DEFAULT_NORMALISATION   # noqa: F401  (names the reference module exports)


def _alphabet(kmer_len):
    """The process-global alphabet (batch.init_chunk_identity_worker); b'ACGT' when no worker initialiser ran."""
    if batch.kmer_alphabet is None:
        from .variables import DEFAULT_ALPHABET
        batch.init_chunk_identity_worker(kmer_len, DEFAULT_ALPHABET)
    return batch.kmer_alphabet


def _as_bytes(seq):
    return seq if isinstance(seq, (bytes, bytearray)) else str(seq).encode('ascii')


# ---------------------------------------------------------------------------------------------------------------------------
# mapping-table helpers (host bookkeeping on the record array)
# ---------------------------------------------------------------------------------------------------------------------------

def convert_mapping_times_to_samples(mapping_table, start_sample, sample_rate):
    """chunkify_raw.py:18-49: a copy of `mapping_table` whose 'start' / 'length' are whole samples counted from the beginning
    of the raw signal (int64) instead of seconds.  Consecutive rows must tile the time axis before and after rounding."""
    descr = [(name, '<i8' if name in ('start', 'length') else kind) for name, kind in mapping_table.dtype.descr]
    start_s, length_s = mapping_table['start'], mapping_table['length']
    assert np.allclose(start_s[:-1] + length_s[:-1], start_s[1:])
    starts = np.around(start_s * sample_rate - start_sample).astype(int)
    lengths = np.around(length_s * sample_rate).astype(int)
    assert np.all(starts[:-1] + lengths[:-1] == starts[1:])
    out = mapping_table.copy().astype(descr)
    out['start'] = starts
    out['length'] = lengths
    return out


def trim_signal_and_mapping(signal, mapping_table, start_sample, end_sample):
    """chunkify_raw.py:52-70: samples [start_sample, end_sample) of `signal`, and the rows of the table that overlap them,
    re-based so that the first row starts at sample 0 and the last row ends with the trimmed signal."""
    sig_trim = signal[start_sample:end_sample]
    end_sample = start_sample + len(sig_trim)
    rows = np.arange(len(mapping_table))
    first = int(rows[mapping_table['start'] > start_sample].min()) - 1       # ValueError on an empty selection, as numpy's
    last = int(rows[mapping_table['start'] < end_sample].max()) + 1
    out = mapping_table[first:last].copy()
    out['start'] -= start_sample
    out['start'][0] = 0
    out['length'][0] = out['start'][1]
    out['length'][-1] = len(sig_trim) - out['start'][-1]
    return sig_trim, out


def mapping_table_is_registered(mapped_signal, mapping_table):
    """chunkify_raw.py:73-83: do the rows tile exactly the samples of `mapped_signal`?"""
    start, length, n = mapping_table['start'], mapping_table['length'], len(mapped_signal)
    return bool(start[0] == 0 and start[-1] + length[-1] == n and (start >= 0).all() and (start < n).all()
                and (start[:-1] + length[:-1] == start[1:]).all())


def replace_repeats_with_zero(arr):
    """chunkify_raw.py:139-142 (in place, like the reference)."""
    arr[np.ediff1d(arr, to_begin=1) == 0] = 0
    return arr


def fill_zeros_with_prev(arr):
    """chunkify_raw.py:145-148."""
    return arr[np.maximum.accumulate(np.arange(len(arr)) * (arr != 0))]


def index_of_previous_non_zero(input_array):
    """chunkify_raw.py:151-155."""
    return np.maximum.accumulate(np.arange(len(input_array)) * (input_array > 0))


def _status(dev_status, what):
    bits = int(dev_status.item())
    if bits & 1:
        raise KeyError("%s: a k-mer has a letter outside the alphabet %r" % (what, batch.kmer_alphabet))
    if bits & 2:
        raise KeyError("%s: an interpolated position falls outside the reference sequence" % what)


def labels_from_mapping_table(kmer_array, kmer_len, index_from=1):
    """chunkify_raw.py:117-136: the middle `kmer_len` letters of every k-mer of `kmer_array` (an 'S<k>' array of any shape) as
    state + index_from, int32, same shape."""
    import torch
    from . import device as D
    kmer_array = np.ascontiguousarray(kmer_array)
    old_len = kmer_array.dtype.itemsize
    assert kmer_array.dtype.kind == 'S' and kmer_len <= old_len
    alphabet = _alphabet(kmer_len)
    if kmer_array.size == 0:
        return np.zeros(kmer_array.shape, dtype='i4')
    text = torch.from_numpy(np.frombuffer(kmer_array.tobytes(), dtype=np.uint8).copy()).to(D.device())
    out = torch.empty(kmer_array.size, dtype=torch.int32, device=text.device)
    status = torch.zeros(1, dtype=torch.int32, device=text.device)
    _lib.check(_lib.lib().slk_kmer_labels_i32(text.data_ptr(), kmer_array.size, old_len, kmer_len, alphabet, len(alphabet),
                                              int(index_from), out.data_ptr(), status.data_ptr(), D.stream_ptr()),
               "labels_from_mapping_table")
    _status(status, "labels_from_mapping_table")
    return out.cpu().numpy().reshape(kmer_array.shape)


def _interp_call(mapping_table, att, t, k, want_labels, zero_repeats, downsample=1, nlabel=None):
    """One launch of slk_raw_chunk_labels_interp_i32: (positions int64, labels int32 or None)."""
    import torch
    from . import device as D
    dev = D.device()
    cols = [torch.from_numpy(np.ascontiguousarray(mapping_table[f], dtype=np.int64)).to(dev)
            for f in ('start', 'length', 'seq_pos')]
    map_k = mapping_table['kmer'].dtype.itemsize              # len(mapping_table['kmer'][0]), chunkify_raw.py:96
    forward = att['direction'] == "+"
    anchor = int(att['ref_start'] if forward else att['ref_stop'])
    times = None
    if t is not None:
        times = torch.from_numpy(np.ascontiguousarray(np.atleast_1d(t), dtype=np.float64)).to(dev)
        nlabel = times.numel()
    pos = torch.empty(nlabel, dtype=torch.int64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    labels = ref = None
    alphabet, nref = b"A", 0
    if want_labels:
        alphabet = _alphabet(k)
        ref_bytes = _as_bytes(att['reference'])
        nref = len(ref_bytes)
        ref = torch.from_numpy(np.frombuffer(ref_bytes, dtype=np.uint8).copy()).to(dev)
        labels = torch.empty(nlabel, dtype=torch.int32, device=dev)
    rc = _lib.lib().slk_raw_chunk_labels_interp_i32(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(),
                                                    len(mapping_table), map_k, int(forward), anchor, D.ptr(ref), nref, int(k),
                                                    alphabet, len(alphabet), nlabel, int(downsample), D.ptr(times),
                                                    int(zero_repeats), D.ptr(labels), pos.data_ptr(), status.data_ptr(),
                                                    D.stream_ptr())
    _lib.check(rc, "interpolate_labels")
    if want_labels:
        _status(status, "interpolate_labels")
    return pos, labels


def interpolate_pos(mapping_table, att):
    """chunkify_raw.py:86-105: a function time -> reference position (int64), linear between the mid-times of the mapped
    blocks.  `att`: the mapping attributes 'direction' and 'ref_start' ('+') or 'ref_stop' ('-')."""
    def interp(t, k=5):
        pos, _ = _interp_call(mapping_table, att, t, k, False, False)
        pos = pos.cpu().numpy()
        return pos if np.ndim(t) else pos[0]
    return interp


def interpolate_labels(mapping_table, att):
    """chunkify_raw.py:108-114: a function time -> label (state + 1) of the reference k-mer at the interpolated position.
    `att` additionally holds 'reference'."""
    def interp(t, k=5):
        _, labels = _interp_call(mapping_table, att, t, k, True, False)
        return labels.cpu().numpy().astype(np.int64)
    return interp


# ---------------------------------------------------------------------------------------------------------------------------
# raw_chunkify
# ---------------------------------------------------------------------------------------------------------------------------

def _trimmed_for_chunks(signal, mapping_table, chunk_len):
    """chunkify_raw.py:168-175: the asserts and the trim to a whole number of chunks."""
    assert len(signal) >= chunk_len
    assert mapping_table_is_registered(signal, mapping_table)
    ml = len(signal) // chunk_len
    signal, mapping_table = trim_signal_and_mapping(signal, mapping_table, 0, ml * chunk_len)
    assert mapping_table_is_registered(signal, mapping_table)
    return ml, signal, mapping_table


def raw_chunkify_many(signals, mapping_tables, chunk_len, kmer_len, normalisation, downsample_factor, on_device=False):
    """raw_chunkify without interpolation (chunkify_raw.py:164-210) for a list of reads: the label generation of all reads is
    ONE pair of launches (k-mer states of every block, then slk_raw_chunk_labels_i32), the normalisation one launch per
    read ('per-read') or one for all chunks ('per-chunk').

    Returns a list of (chunks [ml, chunk_len, 1] float32, labels [ml, ceil(chunk_len / downsample_factor)] int32,
    bad [ml, chunk_len] bool) per read -- numpy, or device tensors with on_device=True."""
    import torch
    from . import device as D
    assert normalisation in AVAILABLE_NORMALISATIONS
    nread = len(signals)
    if nread == 0 or len(mapping_tables) != nread:
        raise ValueError("raw_chunkify_many needs one mapping table per read")
    dev = D.device()
    alphabet = _alphabet(kmer_len)
    trimmed = [_trimmed_for_chunks(s, m, chunk_len) for s, m in zip(signals, mapping_tables)]
    mls = np.asarray([t[0] for t in trimmed], dtype=np.int64)
    nblk = len(range(0, chunk_len, downsample_factor))
    nev = np.asarray([len(t[2]) for t in trimmed], dtype=np.int64)
    ev_off = np.concatenate([[0], np.cumsum(nev)]).astype(np.int64)
    lab_off = np.concatenate([[0], np.cumsum(mls * nblk)]).astype(np.int64)
    old_len = trimmed[0][2]['kmer'].dtype.itemsize
    if any(t[2]['kmer'].dtype.itemsize != old_len for t in trimmed):
        raise ValueError("the mapping tables hold k-mers of different lengths")
    assert kmer_len <= old_len

    def column(name):
        return torch.from_numpy(np.concatenate([np.asarray(t[2][name], dtype=np.int64) for t in trimmed])).to(dev)
    start, move = column('start'), column('move')
    text = b''.join(np.ascontiguousarray(t[2]['kmer']).tobytes() for t in trimmed)
    text = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).to(dev)
    ev_off_d, mls_d, lab_off_d = (torch.from_numpy(a).to(dev) for a in (ev_off, mls, lab_off[:-1].copy()))
    L = _lib.lib()
    total_ev = int(ev_off[-1])
    event_label = torch.empty(total_ev, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.slk_kmer_labels_i32(text.data_ptr(), total_ev, old_len, kmer_len, alphabet, len(alphabet), 1,
                                     event_label.data_ptr(), status.data_ptr(), D.stream_ptr()), "raw_chunkify.kmer_labels")
    nbytes = L.slk_raw_chunk_labels_workspace_bytes(total_ev)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    labels = torch.empty(int(lab_off[-1]), dtype=torch.int32, device=dev)
    _lib.check(L.slk_raw_chunk_labels_i32(start.data_ptr(), move.data_ptr(), event_label.data_ptr(), ev_off_d.data_ptr(), nread,
                                          mls_d.data_ptr(), lab_off_d.data_ptr(), int(mls.max()), int(chunk_len),
                                          int(downsample_factor), ws.data_ptr(), nbytes, labels.data_ptr(), D.stream_ptr()),
               "raw_chunkify.labels")
    # chunks: chunkify_raw.py:176-185
    if normalisation == 'per-chunk':
        allchunks = torch.cat([D.to_dev(t[1]).reshape(int(t[0]), chunk_len) for t in trimmed], dim=0)
        allchunks = batch.normalise_chunks(allchunks, 'per-chunk')
        bounds = np.concatenate([[0], np.cumsum(mls)])
        chunk_sets = [allchunks[int(bounds[i]):int(bounds[i + 1])] for i in range(nread)]
    else:
        chunk_sets = [batch.normalise_chunks(D.to_dev(t[1]).reshape(int(t[0]), chunk_len), normalisation) for t in trimmed]
    _status(status, "raw_chunkify")
    out = []
    for i in range(nread):
        lab = labels[int(lab_off[i]):int(lab_off[i + 1])].reshape(int(mls[i]), nblk)
        chunks = chunk_sets[i][:, :, None]
        bad = torch.zeros((int(mls[i]), chunk_len), dtype=torch.bool, device=dev)       # chunkify_raw.py:206-207
        if not on_device:
            chunks, lab, bad = chunks.cpu().numpy(), lab.cpu().numpy(), bad.cpu().numpy()
        out.append((chunks, lab, bad))
    return out


def raw_chunkify(signal, mapping_table, chunk_len, kmer_len, normalisation, downsample_factor, interpolation,
                 mapping_attrs=None):
    """Labelled chunks of one read (chunkify_raw.py:164-210): (chunks [ml, chunk_len, 1] float32, labels int32
    [ml, ceil(chunk_len / downsample_factor)], bad [ml, chunk_len] bool -- always False for raw models)."""
    if not interpolation:
        return raw_chunkify_many([signal], [mapping_table], chunk_len, kmer_len, normalisation, downsample_factor)[0]
    from . import device as D
    assert normalisation in AVAILABLE_NORMALISATIONS
    ml, signal, mapping_table = _trimmed_for_chunks(signal, mapping_table, chunk_len)
    chunks = batch.normalise_chunks(D.to_dev(signal).reshape(ml, chunk_len), normalisation)
    nlabel = len(range(0, ml * chunk_len, downsample_factor))
    _, labels = _interp_call(mapping_table, mapping_attrs, None, kmer_len, True, True, downsample_factor, nlabel)
    labels = labels.cpu().numpy().astype(np.int64).reshape((ml, -1))      # chunkify_raw.py:189-192 (int64 there: np.array(...) + 1)
    return chunks[:, :, None].cpu().numpy(), labels, np.zeros((ml, chunk_len), dtype=bool)


def raw_chunk_worker(fn, chunk_len, kmer_len, min_length, trim, normalisation, downsample_factor, interpolation=False):
    """Worker of `chunkify raw_identity` for one single-read fast5 file with a stored mapping (chunkify_raw.py:213-257): same
    arguments, the same tuple (chunks, labels, bad) or None with the reference's message on stderr.  `fn` may also be an object
    that offers what the reference reads through untangled.fast5 -- `get_any_mapping_data('template')`, `get_read(raw=True)`,
    `sample_rate`, `start_time` -- since that package is a dependency of the reference, not part of it: sloika_amd.fast5.Fast5
    reads signal and stored basecalls only, so for a plain path the mapping data is reported as unavailable the way the reference
    reports a file without it."""
    try:
        f5 = fn if hasattr(fn, 'get_any_mapping_data') else None
        if f5 is None:
            from . import fast5
            f5 = fast5.Fast5(fn)
        mapping_table, att = f5.get_any_mapping_data('template')
        sig = f5.get_read(raw=True)
        sample_rate = f5.sample_rate
        start_sample = f5.start_time
    except Exception as e:
        sys.stderr.write('Failed to get mapping data from {}.\n{}\n'.format(fn, repr(e)))
        return None
    mapping_table = convert_mapping_times_to_samples(mapping_table, start_sample, sample_rate)
    map_start = mapping_table['start'][0] + trim[0]
    map_end = mapping_table['start'][-1] + mapping_table['length'][-1] - trim[1]
    mapped_signal, mapping_table = trim_signal_and_mapping(sig, mapping_table, map_start, map_end)
    try:
        assert mapping_table_is_registered(mapped_signal, mapping_table)
    except Exception as e:
        sys.stderr.write('Failed to properly register raw signal and mapping table in {}.\n{}\n'.format(fn, repr(e)))
        return None
    if len(mapped_signal) < max(chunk_len, min_length):
        sys.stderr.write('{} is too short.\n'.format(fn))
        return None
    chunks, labels, bad = raw_chunkify(mapped_signal, mapping_table, chunk_len, kmer_len, normalisation, downsample_factor,
                                       interpolation, att)
    return np.ascontiguousarray(chunks), np.ascontiguousarray(labels), np.ascontiguousarray(bad)


# ---------------------------------------------------------------------------------------------------------------------------
# raw_remap
# ---------------------------------------------------------------------------------------------------------------------------

MAPPING_FIELDS = (('start', '<i8'), ('length', '<i8'), ('seq_pos', '<i8'), ('move', '<i8'), ('kmer', None),
                  ('good_emission', '?'))


def _reference_states(ref, kmer_len):
    """k-mers of the reference and their states + 1 (chunkify_raw.py:268-269)."""
    alphabet = _alphabet(kmer_len)
    kmers = np.array(bio.seq_to_kmers(_as_bytes(ref), kmer_len))
    if len(kmers) == 0:
        raise ValueError("the reference is shorter than one k-mer")
    states, _, _ = bio._states(list(kmers), alphabet.decode('ascii'))
    return kmers, [int(s) + 1 for s in states]


def _mapping_table_of_path(path, kmers, nsample, kmer_len, signal):
    """chunkify_raw.py:277-294: one row per network output step."""
    nstep = len(path)
    dtype = [(name, kind if kind else 'S{}'.format(kmer_len)) for name, kind in MAPPING_FIELDS]
    table = np.zeros(nstep, dtype=dtype)
    stride = int(np.ceil(nsample / float(nstep)))
    table['start'] = np.arange(0, nsample, stride, dtype=np.int64) - stride // 2     # raises if the lengths disagree
    table['length'] = stride
    table['seq_pos'] = path
    table['move'] = np.ediff1d(path, to_begin=1)
    table['kmer'] = kmers[path]
    table['good_emission'] = True
    _, table = trim_signal_and_mapping(signal, table, 0, nsample)
    return table


def _posterior_of_read(signal, min_prob, calc_post):
    """chunkify_raw.py:264-266: per-read median/MAD normalisation, the network on [T,1,1], prepare_post -> device [T', S]."""
    from . import device as D
    sig = D.to_dev(np.ascontiguousarray(signal, dtype=np.float32) if not hasattr(signal, 'device') else signal)
    inmat = batch.normalise_chunks(sig.reshape(1, -1), 'per-chunk', out_layout='network')
    return decode.prepare_post(D.to_dev(calc_post(inmat)), min_prob=min_prob, drop_bad=False)


def _calc_post(calc_post):
    f = calc_post if calc_post is not None else batch.calc_post
    if f is None:
        raise ValueError("raw_remap needs a compiled model: pass calc_post= or call batch.init_chunk_remap_worker first")
    return f


def raw_remap_many(refs, signals, min_prob, kmer_len, prior, slip, calc_post=None):
    """raw_remap (chunkify_raw.py:260-296) for a list of reads.  The network runs read by read (batch 1, as the reference's
    worker does); the remap DP of ALL reads is one launch (slk_map_to_sequence_batch_f32, one workgroup per read).
    Returns a list of (score, mapping_table, path, seq)."""
    f = _calc_post(calc_post)
    if len(refs) != len(signals) or len(refs) == 0:
        raise ValueError("raw_remap_many needs one reference per read")
    posts = [_posterior_of_read(s, min_prob, f) for s in signals]
    refk = [_reference_states(r, kmer_len) for r in refs]
    seqs = [s for _, s in refk]
    p0 = None if prior[0] is None else [util.geometric_prior(len(s), prior[0]) for s in seqs]
    p1 = None if prior[1] is None else [util.geometric_prior(len(s), prior[1], rev=True) for s in seqs]
    scores, paths = transducer.map_to_sequence_batch(posts, seqs, slip, prior_initial=p0, prior_final=p1, log=False)
    out = []
    for i, (signal, (kmers, seq)) in enumerate(zip(signals, refk)):
        path = paths[i].astype(np.int64)
        table = _mapping_table_of_path(path, kmers, len(signal), kmer_len, signal)
        out.append((scores[i], table, path, seq))
    return out


def raw_remap(ref, signal, min_prob, kmer_len, prior, slip, calc_post=None):
    """Map raw signal to its reference sequence with the transducer model (chunkify_raw.py:260-296):
    (score float32, mapping_table, path int64[T'], seq = states + 1 of the reference's k-mers).

    `prior` = (mean of the geometric start prior or None, the same for the end); `calc_post` is the compiled model (the
    reference keeps it in the process global `batch.calc_post`)."""
    f = _calc_post(calc_post)
    post = _posterior_of_read(signal, min_prob, f)
    kmers, seq = _reference_states(ref, kmer_len)
    prior0 = None if prior[0] is None else util.geometric_prior(len(seq), prior[0])
    prior1 = None if prior[1] is None else util.geometric_prior(len(seq), prior[1], rev=True)
    score, path = transducer.map_to_sequence(post, seq, slip=slip, prior_initial=prior0, prior_final=prior1, log=False)
    path = path.astype(np.int64)
    return score, _mapping_table_of_path(path, kmers, len(signal), kmer_len, signal), path, seq


def raw_chunk_remap_worker(fn, trim, min_prob, kmer_len, min_length, prior, slip, chunk_len, normalisation, downsample_factor,
                           interpolation, open_pore_fraction, references, calc_post=None):
    """Worker of `chunkify raw_remap` for one single-read fast5 file (chunkify_raw.py:299-337): same arguments, and the same
    tuple (file name, score, rows, path, seq, chunks, labels, bad) or None with a message on stderr."""
    import os
    from . import fast5
    try:
        signal = fast5.Fast5(fn).get_read(raw=True)
        sn = os.path.splitext(os.path.basename(fn))[0]
    except Exception as e:
        sys.stderr.write('Failure reading events from {}.\n{}\n'.format(fn, repr(e)))
        return None
    try:
        read_ref = references[sn]
    except Exception as e:
        sys.stderr.write('No reference found for {}.\n{}\n'.format(fn, repr(e)))
        return None
    signal = batch.trim_open_pore(signal, open_pore_fraction)
    signal = util.trim_array(signal, *trim)
    if len(signal) < max(chunk_len, min_length):
        sys.stderr.write('{} is too short.\n'.format(fn))
        return None
    try:
        score, mapping_table, path, seq = raw_remap(read_ref, signal, min_prob, kmer_len, prior, slip, calc_post=calc_post)
    except Exception as e:
        sys.stderr.write("Failure remapping read {}.\n{}\n".format(sn, repr(e)))
        return None
    attrs = {'reference': read_ref, 'direction': '+', 'ref_start': 0}              # chunkify_raw.py:329-333
    chunks, labels, bad_ev = raw_chunkify(signal, mapping_table, chunk_len, kmer_len, normalisation, downsample_factor,
                                          interpolation, attrs)
    return sn + '.fast5', score, len(mapping_table), path, seq, chunks, labels, bad_ev
