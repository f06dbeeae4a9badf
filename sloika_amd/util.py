"""The two helpers of sloika/util.py that sit on the path (behaviour of util.py:12-26 and :94-99, restated)."""
import numpy as np


def geometric_prior(n, m, rev=False):
    """log P(start at step k), k = 0 .. n-1, for a geometric start time with mean `m` steps: log p + k log(1 - p), p = 1 / (1 + m);
    `rev` gives the same numbers back to front (the prior on the END of the sequence, transducer.py:39-41, 63-64)."""
    p = 1.0 / (1.0 + m)
    steps = np.arange(n, dtype=np.float64)
    steps = steps[::-1] if rev else steps
    out = np.full(n, np.log(p))
    out[steps > 0] += steps[steps > 0] * np.log1p(-p)         # (step 0 keeps log p itself: m = 0 gives p = 1, and 0 * log 0 is not 0)
    return out


def trim_array(x, from_start, from_end):
    """`x` without its first `from_start` and last `from_end` entries (a view); negative counts are refused."""
    if from_start < 0 or from_end < 0:
        raise AssertionError("trim counts must not be negative")
    return x[from_start:len(x) - from_end]
