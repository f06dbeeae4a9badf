"""Constants of the k-mer state space: sloika/variables.py:1-26."""
DEFAULT_ALPHABET = b'ACGT'
DEFAULT_NBASE = len(DEFAULT_ALPHABET)


def nkmer(kmer, nbase=DEFAULT_NBASE):
    """Number of possible kmers of a given length (variables.py:5-13)."""
    return nbase ** kmer


def nstate(kmer, transducer=True, bad_state=True, nbase=DEFAULT_NBASE):
    """Number of states in model (variables.py:16-26)."""
    return nkmer(kmer, nbase=nbase) + (transducer or bad_state)
