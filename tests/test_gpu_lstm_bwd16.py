"""csrc/lstm_bwd16.hip -- the Lstm reverse scan on the barrier-stepped fp16-split plan -- against the float32 FMA kernel of
csrc/train.hip (slk_lstm_backward_f32, itself pinned to the float64 training oracle by tests/test_gpu_train.py), through the C ABI.
Gradients have no natural range: the inputs are scaled from 1e-9 to 1e+3."""
import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu


def _inputs(rs, T, B, n, scale):
    dy = (rs.normal(size=(T * B, n)) * scale).astype(np.float32)
    dy[rs.uniform(size=(T * B, n)) < 0.3] = 0.0                       # steps without loss (train_network.py drops the chunk ends)
    g = np.tanh(rs.normal(size=(T * B, n)))
    gates = np.stack([g, 1 / (1 + np.exp(-rs.normal(size=(T * B, n)))), 1 / (1 + np.exp(-rs.normal(size=(T * B, n)) - 1)),
                      1 / (1 + np.exp(-rs.normal(size=(T * B, n))))], axis=2).reshape(T * B, 4 * n).astype(np.float32)
    cell = (rs.normal(size=(T * B, n)) * 1.5).astype(np.float32)
    sW = (2.0 * rs.normal(size=(4 * n, n)) / np.sqrt(2 * n)).astype(np.float32)
    peep = (rs.normal(size=(3, n)) / np.sqrt(n)).astype(np.float32)
    return dy, gates, cell, sW, peep


def _run(L, entry, dy, gates, cell, sW, peep, T, B, n, reverse):
    import torch
    dsum = torch.full((T * B, 4 * n), float("nan"), device="cuda")
    dpeep = torch.full((B, 3 * n), float("nan"), device="cuda")
    rc = getattr(L, entry)(dy.data_ptr(), n, gates.data_ptr(), cell.data_ptr(), sW.data_ptr(), None if peep is None else peep.data_ptr(),
                           dsum.data_ptr(), dpeep.data_ptr(), T, B, n, int(reverse), 1, 2, stream())
    return rc, dsum, dpeep


@pytest.mark.parametrize("n", [16, 32, 48, 64])
@pytest.mark.parametrize("T,B,reverse,scale", [(23, 9, False, 1.0), (8, 4, True, 1e-9), (3, 2, False, 1e3), (1, 1, True, 1.0),
                                               (61, 5, True, 1e-4), (200, 33, False, 1e-2)])
def test_lstm_bwd16_vs_fp32_kernel(n, T, B, reverse, scale):
    need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(n + T)
    dy, gates, cell, sW, peep = (dev(a) for a in _inputs(rs, T, B, n, scale))
    rc0, want, wantp = _run(L, "slk_lstm_backward_f32", dy, gates, cell, sW, peep, T, B, n, reverse)
    rc1, got, gotp = _run(L, "slk_lstm_backward16_f32", dy, gates, cell, sW, peep, T, B, n, reverse)
    assert rc0 == 0 and rc1 == 0
    # relative to each chunk's largest gradient at that step (what the column scaling of the kernel preserves), and overall
    w, g = want.cpu().numpy().reshape(T, B, 4 * n), got.cpu().numpy().reshape(T, B, 4 * n)
    assert np.isfinite(g).all()
    top = max(float(np.abs(w).max()), 1e-30)
    assert np.abs(g - w).max() <= 3e-5 * top
    assert np.abs(gotp.cpu().numpy() - wantp.cpu().numpy()).max() <= 3e-5 * max(float(np.abs(wantp.cpu().numpy()).max()), 1e-30)


def test_lstm_bwd16_without_peepholes_and_repeats_bit_for_bit():
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    n, T, B = 64, 120, 517
    rs = np.random.RandomState(3)
    dy, gates, cell, sW, peep = (dev(a) for a in _inputs(rs, T, B, n, 1e-3))
    _, want, wantp = _run(L, "slk_lstm_backward_f32", dy, gates, cell, sW, None, T, B, n, True)
    first = None
    for rep in range(3):
        rc, got, gotp = _run(L, "slk_lstm_backward16_f32", dy, gates, cell, sW, None, T, B, n, True)
        assert rc == 0
        if first is None:
            first = (got, gotp)
        else:
            assert torch.equal(first[0], got) and torch.equal(first[1], gotp)
    top = float(want.abs().max())
    assert float((first[0] - want).abs().max()) <= 3e-5 * top


def test_lstm_bwd16_unsupported_shapes_are_refused():
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    z = torch.zeros(4096, device="cuda")
    for n, act, gate in [(96, 1, 2), (24, 1, 2), (64, 2, 2), (64, 1, 1)]:
        assert L.slk_lstm_backward16_f32(z.data_ptr(), n, z.data_ptr(), z.data_ptr(), z.data_ptr(), None, z.data_ptr(), z.data_ptr(),
                                         1, 1, n, 0, act, gate, stream()) == _lib.SLK_ERR_UNSUPPORTED
