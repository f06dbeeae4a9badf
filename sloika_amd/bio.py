"""States -> k-mers -> bases (the step after the decoded path; reference: sloika/bio.py:12-36, 145-237).

A k-mer over an alphabet of `nbase` letters is handled as the base-`nbase` NUMBER of its letters, first letter most
significant -- the index `all_kmers` gives it (bio.py:12-24).  The reference's string tests then become integer ones:

    k1[i:] == k2[:-i]      <=>      s1 mod nbase^(k-i) == s2 div nbase^i

`paths_to_bases` runs whole batches of decoded paths on the device (csrc/bases.hip, C ABI slk_paths_to_bases);
`kmers_to_sequence` & co. keep the reference's string-level signatures for single calls and evaluate the same integer
rule with numpy (they are host-side string utilities in the reference too).
"""
from itertools import product

import numpy as np


def all_kmers(length, alphabet='ACGT'):
    """Every k-mer of `length` letters in alphabet order (bio.py:12-24); bytes in, bytes out."""
    as_bytes = isinstance(alphabet, bytes)
    letters = alphabet.decode('utf-8') if as_bytes else alphabet
    words = (''.join(w) for w in product(letters, repeat=length))
    return [w.encode('utf-8') for w in words] if as_bytes else list(words)


def kmer_mapping(length, alphabet='ACGT'):
    """k-mer -> its index in all_kmers (bio.py:27-36)."""
    return dict((kmer, idx) for idx, kmer in enumerate(all_kmers(length, alphabet)))


def seq_to_kmers(seq, length):
    """The overlapping k-mers of a string, in order (bio.py:145-157)."""
    return [seq[start:start + length] for start in range(len(seq) - length + 1)]


def _alphabet_of(kmers):
    letters = sorted(set(ch for k in kmers for ch in (k.decode('utf-8') if isinstance(k, bytes) else k)))
    return ''.join(letters)


def _states(kmers, alphabet):
    """k-mer strings -> (state numbers int64, k, nbase) under `alphabet` (letter rank = digit)."""
    k = len(kmers[0])
    rank = np.full(256, -1, dtype=np.int64)
    rank[np.frombuffer(alphabet.encode('utf-8'), dtype=np.uint8)] = np.arange(len(alphabet))
    text = b''.join(x if isinstance(x, bytes) else x.encode('utf-8') for x in kmers)
    digits = rank[np.frombuffer(text, dtype=np.uint8)].reshape(len(kmers), k)
    if (digits < 0).any():
        raise ValueError("k-mer letter outside the alphabet %r" % alphabet)
    weights = len(alphabet) ** np.arange(k - 1, -1, -1, dtype=np.int64)
    return digits @ weights, k, len(alphabet)


def moves_of_states(states, k, nbase, allow_identical=True):
    """Moves between consecutive states: 0 for a repeat (when allowed), else the smallest shift i in 1..k-1 whose
    suffix/prefix digits agree, else k (bio.py:160-179)."""
    states = np.asarray(states, dtype=np.int64)
    s1, s2 = states[:-1], states[1:]
    move = np.full(len(s1), k, dtype=np.int64)
    for shift in range(k - 1, 0, -1):                       # later assignments are smaller shifts: the smallest wins
        agree = (s1 % nbase ** (k - shift)) == (s2 // nbase ** shift)
        move[agree] = shift
    if allow_identical:
        move[s1 == s2] = 0
    return move


def sequence_of_states(states, k, nbase, moves, alphabet):
    """First state in full, then the last min(move, k) letters of every following state (bio.py:206-225)."""
    states = np.asarray(states, dtype=np.int64)
    moves = np.minimum(np.asarray(moves, dtype=np.int64), k)
    take = np.concatenate(([k], moves))                      # letters contributed by each state
    total = int(take.sum())
    owner = np.repeat(np.arange(len(states)), take)          # which state each output letter comes from
    start = np.cumsum(take) - take
    within = np.arange(total) - start[owner]                 # 0 .. take-1 inside the contribution
    place = take[owner] - 1 - within                         # digit position counted from the least significant
    digits = (states[owner] // nbase ** place) % nbase
    return np.frombuffer(alphabet.encode('utf-8'), dtype=np.uint8)[digits].tobytes().decode('utf-8')


def max_overlap(kmers, allow_identical=True):
    """List of moves between consecutive k-mer strings (bio.py:160-179)."""
    kmers = list(kmers)
    if len(kmers) < 2:
        return []
    alphabet = _alphabet_of(kmers)
    states, k, nbase = _states(kmers, alphabet)
    return [int(m) for m in moves_of_states(states, k, max(nbase, 2), allow_identical)]


def moves_compatible(kmers, moves):
    """Per transition: does the claimed move agree with the two k-mers (bio.py:182-203)?  A move of k or more always
    does (nothing left to compare), a move of 0 needs identical k-mers."""
    kmers = list(kmers)
    if len(kmers) < 2:
        return []
    alphabet = _alphabet_of(kmers)
    states, k, nbase = _states(kmers, alphabet)
    nbase = max(nbase, 2)
    out = []
    for s1, s2, m in zip(states[:-1], states[1:], moves):
        if m == 0:
            out.append(bool(s1 == s2))
        elif m >= k:
            out.append(True)
        else:
            out.append(bool(s1 % nbase ** (k - m) == s2 // nbase ** m))
    return out


def reduce_kmers(kmers, moves):
    """k-mer strings + moves -> sequence (bio.py:206-225)."""
    kmers = list(kmers)
    assert all(moves_compatible(kmers, moves)), 'Moves not consistent with kmers'
    alphabet = _alphabet_of(kmers)
    states, k, nbase = _states(kmers, alphabet)
    seq = sequence_of_states(states, k, max(nbase, 2), list(moves)[:len(kmers) - 1], alphabet)
    return seq.encode('utf-8') if isinstance(kmers[0], bytes) else seq


def kmers_to_sequence(kmers, always_move=False):
    """Sequence of a list of k-mer strings by maximum overlap (bio.py:228-237)."""
    kmers = list(kmers)
    return reduce_kmers(kmers, max_overlap(kmers, not always_move))


def states_to_sequence(states, klen, alphabet='ACGT', always_move=False):
    """The same for a path of STATE NUMBERS (what decode.viterbi returns): no strings are built on the way."""
    if isinstance(alphabet, bytes):
        alphabet = alphabet.decode('utf-8')
    states = np.asarray(states, dtype=np.int64)
    if states.size == 0:
        return ''
    moves = moves_of_states(states, klen, len(alphabet), not always_move)
    return sequence_of_states(states, klen, len(alphabet), moves, alphabet)


def paths_to_bases(paths, lens, klen, alphabet='ACGT', always_move=True):
    """Batch of decoded paths on the device -> list of base strings.  paths:[B, T] int32 device tensor (rows valid for
    lens[b] entries, as pipeline.Basecaller.call_chunks returns them), lens:[B] int32 device tensor."""
    import torch
    from . import _lib, device as D
    _lib.require_gpu()
    if isinstance(alphabet, str):
        alphabet = alphabet.encode('utf-8')
    B, Tmax = paths.shape
    cap = max(klen * max(Tmax, 1), klen)
    out = torch.empty((B, cap), dtype=torch.uint8, device=paths.device)
    nb = torch.empty((B,), dtype=torch.int32, device=paths.device)
    packed = int.from_bytes(alphabet.ljust(8, b'\0'), 'little')
    _lib.check(_lib.lib().slk_paths_to_bases(paths.data_ptr(), paths.stride(0), lens.data_ptr(), B, klen, len(alphabet),
                                             int(bool(always_move)), packed, out.data_ptr(), cap, nb.data_ptr(),
                                             D.stream_ptr()), "paths_to_bases")
    counts = nb.cpu().numpy()
    host = out[:, :int(counts.max()) if B else 0].cpu().numpy()
    return [host[b, :counts[b]].tobytes().decode('utf-8') for b in range(B)]
