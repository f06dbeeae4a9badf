"""transducer.map_to_sequence (sloika/transducer.py:14-73) through the C ABI."""
import numpy as np

from . import _lib

_NEG_LARGE = -50000.0
_STAY = 0


def map_to_sequence(trans, sequence, slip=None, prior_initial=None, prior_final=None, log=True):
    """Find Viterbi path through sequence for transducer.

    :param trans: 2D array [nev, nstate], transducer posteriors (log scaled if `log`)
    :param sequence: 1D array of state indices to be mapped against
    :param slip: slip penalty (in log-space)
    :param prior_initial / prior_final: 1D float64 arrays, prior over initial / final position
    :returns: (score float32, path int32[nev])
    """
    import torch
    from . import device as D
    assert slip is None or slip >= 0.0, 'Slip penalty should be non-negative'
    if slip is None:
        # transducer.py:27 turns None into float32(nan): every slip comparison is then false and the slip
        # move always wins -- not a usable mode; refuse instead of reproducing it.
        raise ValueError("map_to_sequence needs a slip penalty (the reference's slip=None evaluates to NaN)")
    td = D.to_dev(trans)
    if td.dim() != 2:
        raise ValueError("map_to_sequence expects [time, state]")
    if not log:
        # transducer.py:30: np.log(trans)
        lt = torch.empty_like(td)
        _lib.check(_lib.lib().slk_log_post_f32(td.data_ptr(), lt.data_ptr(), td.numel(), _lib.POST_LN, 0.0, D.stream_ptr()),
                   "map_to_sequence.log")
        td = lt
    nev, nst = td.shape
    seq = torch.as_tensor(np.ascontiguousarray(sequence, dtype=np.int32)).to(td.device)
    npos = seq.shape[0]
    pi = None if prior_initial is None else torch.as_tensor(np.ascontiguousarray(prior_initial, dtype=np.float64)).to(td.device)
    pf = None if prior_final is None else torch.as_tensor(np.ascontiguousarray(prior_final, dtype=np.float64)).to(td.device)
    L = _lib.lib()
    nbytes = L.slk_map_to_sequence_workspace_bytes(nev, npos)
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=td.device)
    score = torch.empty(1, dtype=torch.float32, device=td.device)
    path = torch.empty(nev, dtype=torch.int32, device=td.device)
    rc = L.slk_map_to_sequence_f32(td.data_ptr(), nev, nst, seq.data_ptr(), npos, float(slip), D.ptr(pi), D.ptr(pf),
                                   ws.data_ptr(), nbytes, score.data_ptr(), path.data_ptr(), D.stream_ptr())
    _lib.check(rc, "map_to_sequence")
    return np.float32(score.item()), path.cpu().numpy()
