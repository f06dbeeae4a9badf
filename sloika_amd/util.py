"""The two helpers of sloika/util.py that sit on the path."""
import numpy as np


def geometric_prior(n, m, rev=False):
    """Log probabilities for random start time with geometric distribution (util.py:12-26)."""
    p = 1.0 / (1.0 + m)
    prior = np.repeat(np.log(p), n)
    prior[1:] += np.arange(1, n) * np.log1p(-p)
    if rev:
        prior = prior[::-1]
    return prior


def trim_array(x, from_start, from_end):
    """util.py:94-99"""
    assert from_start >= 0
    assert from_end >= 0
    from_end = None if from_end == 0 else -from_end
    return x[from_start:from_end]
