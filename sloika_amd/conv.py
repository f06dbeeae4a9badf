"""Padding arithmetic of sloika/conv.py:10-63 (host-side; the convolution itself is csrc/frontend.hip)."""

PADDING_MODES = frozenset(['same', 'half', 'valid', 'full', 'same_left'])


def calculate_padding(mode, winlen):
    """Calculate padding amount for given convolution mode and window length (conv.py:10-63)

        'same'        [(winlen - 1) // 2, winlen // 2]      tensorflow 'SAME'
        'half'        [winlen // 2, winlen // 2]            Theano 'half'
        'valid'       [0, 0]
        'full'        [winlen - 1, winlen - 1]
        'same_left'   [winlen // 2, (winlen - 1) // 2]
        int           [int, int]
        (int1, int2)  [int1, int2]

    :returns: (padding to start, padding to end)
    """
    assert winlen > 0, "winlen must be positive"
    if isinstance(mode, int):
        return (mode, mode)
    if isinstance(mode, (tuple, list)):
        # conv.py:47-49 intends this; its py2-era `map(type, mode) == [int, int]` never matches on python 3
        if len(mode) == 2 and all(isinstance(m, int) for m in mode):
            return tuple(mode)

    assert mode in PADDING_MODES, 'Padding mode "{}" not supported'.format(mode)
    if mode == "same":
        return ((winlen - 1) // 2, winlen // 2)
    if mode == "half":
        return (winlen // 2, winlen // 2)
    if mode == "valid":
        return (0, 0)
    if mode == "full":
        return (winlen - 1, winlen - 1)
    if mode == "same_left":
        return (winlen // 2, (winlen - 1) // 2)

    raise NotImplementedError("Padding mode case {} not dealt with".format(mode))
