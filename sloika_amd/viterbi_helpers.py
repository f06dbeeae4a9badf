"""viterbi_helpers.slip_update (sloika/viterbi_helpers.pyx:12-35) through the C ABI."""
import numpy as np

from . import _lib


def slip_update(x, slip):
    """Efficiently compute the score for a geometric slip.

    :param x: 1D float32 array (len >= 3: the reference writes index 2 unconditionally)
    :param slip: slip penalty (log space)
    :returns: (from_score float32[n], from_pos int64[n])
    """
    import torch
    from . import device as D
    xd = D.to_dev(x)
    if xd.dim() != 1:
        raise ValueError("slip_update expects a 1D array")
    n = xd.shape[0]
    fs = torch.empty(n, dtype=torch.float32, device=xd.device)
    fp = torch.empty(n, dtype=torch.int64, device=xd.device)
    _lib.check(_lib.lib().slk_slip_update_f32(xd.data_ptr(), n, float(slip), fs.data_ptr(), fp.data_ptr(),
                                              D.stream_ptr()), "slip_update")
    if isinstance(x, torch.Tensor):
        return fs, fp
    return fs.cpu().numpy(), fp.cpu().numpy()
