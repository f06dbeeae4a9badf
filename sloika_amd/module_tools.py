"""The namespace models/*.py import as `smt` (sloika/module_tools.py:1-13)."""
from functools import partial

from scipy.stats import truncnorm

from .config import sloika_dtype
from .activation import *          # noqa: F401,F403
from .layers import *              # noqa: F401,F403
from .layers import birnn, zeros   # noqa: F401
from .variables import *           # noqa: F401,F403
from .variables import DEFAULT_NBASE, DEFAULT_ALPHABET, nkmer, nstate  # noqa: F401


def truncated_normal(size, sd):
    ''' Truncated normal for Xavier style initiation (module_tools.py:9-13) '''
    res = sd * truncnorm.rvs(-2, 2, size=size)
    return res.astype(sloika_dtype)
