"""Padding arithmetic of the Convolution layer (reference: sloika/conv.py:10-63; the convolution itself is
csrc/frontend.hip).  A mode names how many zero steps go in front of and behind the time axis."""

#: mode -> (front, back) as functions of the window length w
_PADDING = {
    'same': lambda w: ((w - 1) // 2, w // 2),            # output length = ceil(T / stride) (tensorflow 'SAME')
    'half': lambda w: (w // 2, w // 2),                  # Theano 'half'
    'valid': lambda w: (0, 0),
    'full': lambda w: (w - 1, w - 1),
    'same_left': lambda w: (w // 2, (w - 1) // 2),
}
PADDING_MODES = frozenset(_PADDING)


def calculate_padding(mode, winlen):
    """(padding to start, padding to end) for a mode name, a single int (both ends) or an (int, int) pair.

    The pair form is what conv.py:47-49 intends; there the py2-era test `map(type, mode) == [int, int]` never holds on
    python 3, so the reference itself cannot reach it."""
    assert winlen > 0, "winlen must be positive"
    if isinstance(mode, int):
        return (mode, mode)
    if isinstance(mode, (tuple, list)) and len(mode) == 2 and all(isinstance(m, int) for m in mode):
        return tuple(mode)
    assert mode in PADDING_MODES, 'Padding mode "{}" not supported'.format(mode)
    return _PADDING[mode](winlen)
