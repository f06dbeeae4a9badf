"""csrc/lstm_scan16.hip -- the Lstm scan (sloika/layers.py:677-691) on the barrier-stepped fp16-split plan -- through the C ABI,
against the oracle (float32 C port, itself pinned to the reference's layers.py by tests/test_oracle_reference_layers.py)."""
import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu


def _params(rs, I, n, scale=1.0):
    iW = (rs.normal(size=(4 * n, I)) / np.sqrt(I + n)).astype(np.float32)
    sW = (scale * rs.normal(size=(4 * n, n)) / np.sqrt(2 * n)).astype(np.float32)
    b = rs.normal(size=4 * n).astype(np.float32)
    p = (rs.normal(size=(3, n)) / np.sqrt(n)).astype(np.float32)
    return iW, sW, b, p


def _scan(L, vW, sW, p, y, ldy, T, B, n, reverse, lens=None, act=1, gate=2):
    return L.slk_lstm_scan16_f32(vW.data_ptr(), sW.data_ptr(), None if p is None else p.data_ptr(), y.data_ptr(), ldy, T, B, n,
                                 int(reverse), act, gate, None if lens is None else lens.data_ptr(), stream())


@pytest.mark.parametrize("n", [16, 32, 48, 64, 80, 96, 128])
@pytest.mark.parametrize("T,B,reverse,peep", [(23, 9, False, True), (8, 4, True, True), (3, 2, False, False), (1, 1, True, True),
                                              (41, 5, True, False), (100, 33, False, True)])
def test_lstm_scan16_vs_oracle(oracle, n, T, B, reverse, peep):
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    I = 24
    rs = np.random.RandomState(n + T)
    iW, sW, b, p = _params(rs, I, n, scale=2.0)
    if not peep:
        p = np.zeros_like(p)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    ref = oracle.lstm(x, iW, sW, b, p, reverse=reverse)
    vW = (x.reshape(T * B, I).astype(np.float64) @ iW.T.astype(np.float64) + b).astype(np.float32)
    yw = torch.full((T, B, n + 16), np.nan, dtype=torch.float32, device="cuda")     # the output as a slice of a wider tensor (birnn)
    assert _scan(L, dev(vW), dev(sW), dev(p) if peep else None, yw, n + 16, T, B, n, reverse) == 0
    out = yw.cpu().numpy()
    assert np.isnan(out[:, :, n:]).all()
    err = np.abs(out[:, :, :n] - ref).max()
    assert err < 2e-5, err


@pytest.mark.parametrize("n", [32, 64, 96, 128])
def test_lstm_scan16_ragged(oracle, n):
    """Each chunk of a ragged batch equals the call on the chunk alone at its own length, reversed scans included; rows past a
    chunk's end stay untouched."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    I, T = 12, 29
    rs = np.random.RandomState(n)
    lens = [29, 1, 20, 8, 28, 9, 2]
    B = len(lens)
    iW, sW, b, p = _params(rs, I, n, scale=2.0)
    x = np.zeros((T, B, I), dtype=np.float32)
    for bb, tb in enumerate(lens):
        x[:tb, bb] = rs.normal(size=(tb, I))
    vW = dev((x.reshape(T * B, I).astype(np.float64) @ iW.T.astype(np.float64) + b).astype(np.float32))
    ld = dev(np.asarray(lens, dtype=np.int32))
    for reverse in (False, True):
        y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
        assert _scan(L, vW, dev(sW), dev(p), y, n, T, B, n, reverse, lens=ld) == 0
        out = y.cpu().numpy()
        for bb, tb in enumerate(lens):
            want = oracle.lstm(x[:tb, bb:bb + 1], iW, sW, b, p, reverse=reverse)
            np.testing.assert_allclose(out[:tb, bb:bb + 1], want, atol=2e-5, err_msg="chunk %d" % bb)
            assert np.isnan(out[tb:, bb]).all()


@pytest.mark.parametrize("n", [64, 128])
def test_lstm_scan16_large_weights_and_determinism(oracle, n):
    """|w| up to 6 with saturating gates; every launch must reproduce the first bit for bit (waves exchange the state through LDS)."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    I, T, B = 12, 60, 1021
    rs = np.random.RandomState(5)
    iW, sW, b, p = _params(rs, I, n, scale=2.0)
    sW[rs.randint(0, 4 * n, 60), rs.randint(0, n, 60)] = rs.choice([-6.0, 6.0, 4.5], size=60)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    vW = dev((x.reshape(T * B, I).astype(np.float64) @ iW.T.astype(np.float64) + b).astype(np.float32))
    sWd, pd = dev(sW), dev(p)
    first = None
    for rep in range(4):
        y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
        assert _scan(L, vW, sWd, pd, y, n, T, B, n, True) == 0
        if first is None:
            first = y
        else:
            assert torch.equal(first, y)
    pick = [0, 3, 500, 1020]
    ref = oracle.lstm(x[:, pick], iW, sW, b, p, reverse=True)
    assert np.abs(first.cpu().numpy()[:, pick] - ref).max() < 5e-5


def test_lstm_scan16_unsupported_shapes_are_refused():
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    z = torch.zeros(4096, device="cuda")
    for n, act, gate in [(144, 1, 2), (24, 1, 2), (64, 2, 2), (64, 1, 1)]:
        assert _scan(L, z, z, z, z, n, 1, 1, n, 0, act=act, gate=gate) == _lib.SLK_ERR_UNSUPPORTED
    assert L.slk_lstm_scan16_f32(None, z.data_ptr(), z.data_ptr(), z.data_ptr(), 64, 1, 1, 64, 0, 1, 2, None, stream()) == _lib.SLK_ERR_INVALID_ARG
