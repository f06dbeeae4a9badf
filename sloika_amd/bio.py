"""String side of the path: states -> k-mers -> bases (sloika/bio.py:12-36, 145-237).

Host-side string manipulation in the reference too (it runs after the device work, on a few hundred k-mers
per read).  Semantics kept exactly: `max_overlap` returns the SMALLEST shift i with k1[i:] == k2[:-i].
"""
from itertools import product


def all_kmers(length, alphabet='ACGT'):
    """All possible kmers of given length, sorted by the ordering of the alphabet (bio.py:12-24)."""
    if isinstance(alphabet, bytes):
        alphabet = alphabet.decode('utf-8')
        return [''.join(x).encode('utf-8') for x in product(alphabet, repeat=length)]
    return [''.join(x) for x in product(alphabet, repeat=length)]


def kmer_mapping(length, alphabet='ACGT'):
    """Dictionary mapping kmer to lexicographical order (bio.py:27-36)."""
    return {k: i for i, k in enumerate(all_kmers(length, alphabet))}


def seq_to_kmers(seq, length):
    """'ATATGCG' => ['ATA','TAT', 'ATG', 'TGC', 'GCG'] (bio.py:145-157)."""
    return [seq[x:x + length] for x in range(0, len(seq) - length + 1)]


def max_overlap(kmers, allow_identical=True):
    """Maximum overlap from one kmer to the next, as a list of moves (bio.py:160-179)."""
    res = []
    for k1, k2 in zip(kmers, kmers[1:]):
        move = len(k1)
        if allow_identical and k1 == k2:
            move = 0
        else:
            for i in range(1, len(k1)):
                if k1[i:] == k2[:-i]:
                    move = i
                    break
        res.append(move)
    return res


def moves_compatible(kmers, moves):
    """Whether moves are compatible with list of kmers (bio.py:182-203)."""
    res = []
    for (k1, k2), m in zip(zip(kmers, kmers[1:]), moves):
        res.append((m == 0 and k1 == k2) or (k1[m:] == k2[:-m]))
    return res


def reduce_kmers(kmers, moves):
    """Reduce a list of kmers to a sequence given the moves between them (bio.py:206-225)."""
    assert all(moves_compatible(kmers, moves)), 'Moves not consistent with kmers'
    kiter = iter(kmers)
    seq = next(kiter)
    for k, m in zip(kiter, moves):
        if m == 0:
            continue
        if m >= len(k):
            seq += k
            continue
        seq += k[-m:]
    return seq


def kmers_to_sequence(kmers, always_move=False):
    """Produce a sequence from kmers by maximum overlap (bio.py:228-237)."""
    kmers = list(kmers)
    moves = max_overlap(kmers, not always_move)
    return reduce_kmers(kmers, moves)
