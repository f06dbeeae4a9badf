"""CPU restatement of the remap -> mapping table -> chunk labels chain (second half).

TEST INFRASTRUCTURE ONLY (same rule as oracle.py): imported by tests/ -- never by the product package `sloika_amd`.
Pinned by tests/golden/remap.npz, which holds the outputs of the reference's own raw_remap / raw_chunkify run in the build
container (tests/golden/make_remap_goldens.py); tests/test_oracle_remap.py checks every function below against it.

    states_of_reference(ref, k)                       bio.seq_to_kmers + kmer_to_state + 1     chunkify_raw.py:268-269
    geometric_prior(n, m, rev)                        sloika/util.py:12-26
    raw_remap(ref, signal, post, ...)                 sloika/tools/chunkify_raw.py:260-296 (the network output is an input)
    trim_table(cols, nsample, lo, hi)                 trim_signal_and_mapping                  chunkify_raw.py:52-70
    chunk_labels(cols, states, ml, chunk_len, ds)     raw_chunkify, plain branch               chunkify_raw.py:194-204
    chunk_labels_interp(cols, ref, ...)               raw_chunkify, interpolated branch        chunkify_raw.py:86-114, 187-192

A mapping table is handled as a dict of equally long int64 columns ('start', 'length', 'seq_pos', 'move') -- the 'kmer'
column of the reference's record array is redundant with seq_pos for a remapped read (kmer = kmers_of_reference[seq_pos]).
"""
import numpy as np

from . import oracle

_RANK = {65: 0, 67: 1, 71: 2, 84: 3}        # A C G T


def states_of_reference(ref, k=5):
    """State + 1 of every overlapping k-mer of `ref` (bytes): first letter most significant (bio.all_kmers order)."""
    digits = np.asarray([_RANK[c] for c in bytes(ref)], dtype=np.int64)
    n = len(digits) - k + 1
    out = np.zeros(n, dtype=np.int64)
    for j in range(k):
        out = out * 4 + digits[j:j + n]
    return out + 1


def geometric_prior(n, m, rev=False):
    """util.py:12-26: log P(start at position i) for a geometric distribution of mean m, float64."""
    p = 1.0 / (1.0 + m)
    lp = np.full(n, np.log(p))
    lp[1:] += np.arange(1, n) * np.log1p(-p)
    return lp[::-1] if rev else lp


def trim_table(cols, nsample, lo, hi):
    """chunkify_raw.py:52-70 on the columns: rows overlapping samples [lo, min(hi, nsample)), re-based to start at 0."""
    ntrim = len(range(*slice(lo, hi).indices(nsample)))
    hi = lo + ntrim
    start = cols['start']
    first = int(np.flatnonzero(start > lo).min()) - 1
    last = int(np.flatnonzero(start < hi).max()) + 1
    out = {k: v[first:last].copy() for k, v in cols.items()}
    out['start'] -= lo
    out['start'][0] = 0
    out['length'][0] = out['start'][1]
    out['length'][-1] = ntrim - out['start'][-1]
    return out


def raw_remap(ref, signal, post, min_prob, k, prior, slip):
    """chunkify_raw.py:260-296 with the network's output `post` [T', nstate] given: (score, table columns, path, seq).
    The normalisation of the signal only feeds the network, so it does not appear."""
    lp = oracle.prepare_post(post[:, None, :], min_prob)
    seq = states_of_reference(ref, k)
    p0 = None if prior[0] is None else geometric_prior(len(seq), prior[0])
    p1 = None if prior[1] is None else geometric_prior(len(seq), prior[1], rev=True)
    score, path = oracle.map_to_sequence(lp, seq, slip=slip, prior_initial=p0, prior_final=p1, log=False)
    path = np.asarray(path, dtype=np.int64)
    nsample, nstep = len(signal), len(path)
    stride = -(-nsample // nstep)
    start = np.arange(0, nsample, stride, dtype=np.int64) - stride // 2
    assert len(start) == nstep
    move = np.empty(nstep, dtype=np.int64)
    move[0] = 1
    move[1:] = np.diff(path)
    cols = {'start': start, 'length': np.full(nstep, stride, dtype=np.int64), 'seq_pos': path.copy(), 'move': move}
    return score, trim_table(cols, nsample, 0, nsample), path, seq


def chunk_labels(cols, states, ml, chunk_len, ds):
    """chunkify_raw.py:194-204, sample by sample: `states[i]` is the label (state + 1) of row i's k-mer.  -> int32 [ml, nblk]."""
    start, move = cols['start'], cols['move']
    ub = ml * chunk_len
    which = np.zeros(ub, dtype=np.int64)              # 1-based number of the move that labels each sample, 0 = none yet
    count = 0
    moved_label = [0]
    for i in range(len(start)):
        if move[i] > 0:
            count += 1
            moved_label.append(int(states[i]))
            which[start[i]] = count
    for s in range(1, ub):
        if which[s] == 0:
            which[s] = which[s - 1]
    kept = which.reshape(ml, chunk_len)[:, ::ds].copy()
    repeat = np.zeros(kept.shape, dtype=bool)
    repeat[:, 1:] = kept[:, 1:] == kept[:, :-1]
    labels = np.asarray(moved_label, dtype=np.int64)[kept]
    labels[repeat] = 0
    return labels.astype(np.int32)


def interp_positions(cols, times, k, map_k, forward=True, anchor=0):
    """chunkify_raw.py:86-105."""
    mid = cols['start'] + 0.5 * cols['length']
    if forward:
        refpos = cols['seq_pos'] + 0.5 * map_k - anchor
    else:
        refpos = anchor - cols['seq_pos'] + 0.5 * map_k
    return np.around(np.interp(times, mid, refpos) - 0.5 * k + 1e-10).astype(np.int64)


def chunk_labels_interp(cols, ref, ml, chunk_len, ds, k, map_k):
    """chunkify_raw.py:187-192 for a '+' mapping with ref_start 0: -> int32 [ml, nblk]."""
    times = np.arange(0, ml * chunk_len, ds)
    pos = interp_positions(cols, times, k, map_k)
    labels = states_of_reference(ref, k)[pos]
    labels[1:][pos[1:] == pos[:-1]] = 0
    return labels.reshape(ml, -1).astype(np.int32)
