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


def map_to_sequence_batch(trans_list, sequence_list, slip, prior_initial=None, prior_final=None, log=True):
    """`map_to_sequence` for many reads in ONE launch (one workgroup per read; the reference's remap loops over reads,
    bin/chunkify.py).  `trans_list[b]`: [nev_b, nstate]; `sequence_list[b]`: state indices; priors: optional lists of
    float64 arrays (all reads or none).  Returns (scores float32[nread], [path int32[nev_b]])."""
    import torch
    from . import device as D
    assert slip is not None and slip >= 0.0, 'Slip penalty should be non-negative'
    nread = len(trans_list)
    if nread == 0 or len(sequence_list) != nread:
        raise ValueError("map_to_sequence_batch needs one sequence per read")
    tds = [D.to_dev(t) for t in trans_list]
    nst = tds[0].shape[1]
    if any(t.dim() != 2 or t.shape[1] != nst for t in tds):
        raise ValueError("map_to_sequence_batch expects [time, state] arrays over the same states")
    nev = [int(t.shape[0]) for t in tds]
    npos = [len(q) for q in sequence_list]
    if min(npos) < 3 or min(nev) < 1:
        raise ValueError("every read needs at least one event and three sequence positions")
    dev = tds[0].device
    td = torch.cat(tds, dim=0).contiguous()
    L = _lib.lib()
    if not log:
        lt = torch.empty_like(td)
        _lib.check(L.slk_log_post_f32(td.data_ptr(), lt.data_ptr(), td.numel(), _lib.POST_LN, 0.0, D.stream_ptr()),
                   "map_to_sequence.log")
        td = lt
    ev_off = np.concatenate([[0], np.cumsum(nev)]).astype(np.int64)
    pos_off = np.concatenate([[0], np.cumsum(npos)]).astype(np.int64)
    ws_sizes = np.array([e * p for e, p in zip(nev, npos)], dtype=np.int64)
    ws_off = np.concatenate([[0], np.cumsum(ws_sizes)[:-1]]).astype(np.int64)
    seq = torch.as_tensor(np.concatenate([np.asarray(q, dtype=np.int32) for q in sequence_list])).to(dev)

    def cat_prior(pl):
        if pl is None:
            return None
        if len(pl) != nread or any(len(p) != n for p, n in zip(pl, npos)):
            raise ValueError("priors must have one float64 value per sequence position of every read")
        return torch.as_tensor(np.concatenate([np.asarray(p, dtype=np.float64) for p in pl])).to(dev)
    pi, pf = cat_prior(prior_initial), cat_prior(prior_final)
    ev_d, pos_d, wso_d = (torch.as_tensor(a).to(dev) for a in (ev_off, pos_off, ws_off))
    ws = torch.empty(int(ws_sizes.sum()), dtype=torch.int32, device=dev)
    score = torch.empty(nread, dtype=torch.float32, device=dev)
    path = torch.empty(int(ev_off[-1]), dtype=torch.int32, device=dev)
    rc = L.slk_map_to_sequence_batch_f32(td.data_ptr(), nst, ev_d.data_ptr(), seq.data_ptr(), pos_d.data_ptr(), nread,
                                         max(npos), float(slip), D.ptr(pi), D.ptr(pf), ws.data_ptr(), wso_d.data_ptr(),
                                         score.data_ptr(), path.data_ptr(), D.stream_ptr())
    _lib.check(rc, "map_to_sequence_batch")
    ph = path.cpu().numpy()
    return score.cpu().numpy(), [ph[ev_off[b]:ev_off[b + 1]] for b in range(nread)]
