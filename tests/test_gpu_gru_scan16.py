"""csrc/gru_scan16.hip -- the Gru scan for layers too wide for the fused kernels (n = 112 / 128 / 144: models/pretrained.pkl,
raw_1.00_rGr zero-padded) on the barrier-stepped fp16-split plan -- through the C ABI, against the oracle (float32 C port,
itself pinned to the reference's layers.py by tests/test_oracle_reference_layers.py)."""
import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu


def _params(rs, I, n, scale=1.0):
    iW = (rs.normal(size=(3 * n, I)) / np.sqrt(I + n)).astype(np.float32)
    sW = (scale * rs.normal(size=(2 * n, n)) / np.sqrt(2 * n)).astype(np.float32)
    sW2 = (scale * rs.normal(size=(n, n)) / np.sqrt(2 * n)).astype(np.float32)
    b = rs.normal(size=3 * n).astype(np.float32)
    return iW, sW, sW2, b


def _scan(L, vI, ldv, sW, sW2, y, ldy, T, B, n, reverse, lens=None):
    return L.slk_gru_scan16_f32(vI.data_ptr(), ldv, sW.data_ptr(), sW2.data_ptr(), y.data_ptr(), ldy, T, B, n, int(reverse), 1, 2,
                                None if lens is None else lens.data_ptr(), stream())


@pytest.mark.parametrize("n", [112, 128, 144])
@pytest.mark.parametrize("T,B,reverse", [(23, 9, False), (8, 4, True), (3, 2, False), (1, 1, True), (41, 5, True), (100, 33, False)])
def test_scan16_vs_oracle(oracle, n, T, B, reverse):
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    I = 48
    rs = np.random.RandomState(n + T)
    iW, sW, sW2, b = _params(rs, I, n, scale=2.0)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    ref = oracle.gru(x, iW, sW, sW2, b, reverse=reverse)
    vI = (x.reshape(T * B, I).astype(np.float64) @ iW.T.astype(np.float64) + b).astype(np.float32)
    # the projection as a slice of a wider workspace, the output as a slice of a concatenated tensor (as birnn produces them)
    ws = torch.zeros((T * B, 3 * n + 8), device="cuda")
    ws[:, :3 * n] = dev(vI)
    yw = torch.full((T, B, n + 16), np.nan, dtype=torch.float32, device="cuda")
    assert _scan(L, ws, 3 * n + 8, dev(sW), dev(sW2), yw, n + 16, T, B, n, reverse) == 0
    out = yw.cpu().numpy()
    assert np.isnan(out[:, :, n:]).all()
    err = np.abs(out[:, :, :n] - ref).max()
    assert err < 2e-5, err


@pytest.mark.parametrize("n", [112, 128, 144])
def test_scan16_ragged(oracle, n):
    """Each chunk of a ragged batch must equal the call on the chunk alone at its own length, reversed scans included; rows past
    a chunk's end stay untouched."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    I = 32
    rs = np.random.RandomState(n)
    T = 29
    lens = [29, 1, 20, 8, 28, 9, 2]
    B = len(lens)
    iW, sW, sW2, b = _params(rs, I, n, scale=2.0)
    x = np.zeros((T, B, I), dtype=np.float32)
    for bb, tb in enumerate(lens):
        x[:tb, bb] = rs.normal(size=(tb, I))
    vI = dev((x.reshape(T * B, I).astype(np.float64) @ iW.T.astype(np.float64) + b).astype(np.float32))
    ld = dev(np.asarray(lens, dtype=np.int32))
    for reverse in (False, True):
        y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
        assert _scan(L, vI, 3 * n, dev(sW), dev(sW2), y, n, T, B, n, reverse, lens=ld) == 0
        out = y.cpu().numpy()
        for bb, tb in enumerate(lens):
            want = oracle.gru(x[:tb, bb:bb + 1], iW, sW, sW2, b, reverse=reverse)
            np.testing.assert_allclose(out[:tb, bb:bb + 1], want, atol=2e-5, err_msg="chunk %d" % bb)
            assert np.isnan(out[tb:, bb]).all()


@pytest.mark.parametrize("n", [112, 144])
def test_scan16_weights_as_trained_and_determinism(oracle, n):
    """|w| up to 6 with saturating gates, as in models/pretrained.pkl; and a kernel whose waves exchange data through LDS without
    enough ordering gives different bits from launch to launch: every launch must reproduce the first."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    I, T, B = 128, 40, 1021
    rs = np.random.RandomState(5)
    iW, sW, sW2, b = _params(rs, I, n, scale=2.0)
    sW2[rs.randint(0, n, 40), rs.randint(0, n, 40)] = rs.choice([-6.0, 6.0, 4.5], size=40)
    sW[rs.randint(0, 2 * n, 40), rs.randint(0, n, 40)] = rs.choice([-5.0, 5.5], size=40)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    vI = dev((x.reshape(T * B, I).astype(np.float64) @ iW.T.astype(np.float64) + b).astype(np.float32))
    sWd, sW2d = dev(sW), dev(sW2)
    first = None
    for rep in range(4):
        y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
        assert _scan(L, vI, 3 * n, sWd, sW2d, y, n, T, B, n, True) == 0
        if first is None:
            first = y
        else:
            assert torch.equal(first, y)
    pick = [0, 3, 500, 1020]
    ref = oracle.gru(x[:, pick], iW, sW, sW2, b, reverse=True)
    assert np.abs(first.cpu().numpy()[:, pick] - ref).max() < 5e-5


def test_scan16_unsupported_shapes_are_refused():
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    z = torch.zeros(4096, device="cuda")
    for n, act, gate in [(160, 1, 2), (96, 1, 2), (120, 1, 2), (128, 2, 2), (144, 1, 1)]:
        assert L.slk_gru_scan16_f32(z.data_ptr(), 3 * n, z.data_ptr(), z.data_ptr(), z.data_ptr(), n, 1, 1, n, 0, act, gate, None,
                                    stream()) == _lib.SLK_ERR_UNSUPPORTED
    assert L.slk_gru_scan16_f32(None, 384, z.data_ptr(), z.data_ptr(), z.data_ptr(), 128, 1, 1, 128, 0, 1, 2, None, stream()) == _lib.SLK_ERR_INVALID_ARG
