"""csrc/lstm_fused16.hip -- a whole Lstm layer (sloika/layers.py:677-697: projection + scan) in one kernel -- through the C ABI,
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


def _layer(L, x, ldx, iW, sW, b, p, y, ldy, T, B, I, n, reverse, lens=None, act=1, gate=2):
    return L.slk_lstm_fused16_f32(x.data_ptr(), ldx, iW.data_ptr(), sW.data_ptr(), None if b is None else b.data_ptr(),
                                  None if p is None else p.data_ptr(), y.data_ptr(), ldy, T, B, I, n, int(reverse), act, gate,
                                  None if lens is None else lens.data_ptr(), stream())


@pytest.mark.parametrize("n", [16, 32, 48, 64])
@pytest.mark.parametrize("I", [4, 12, 32, 36, 64])
@pytest.mark.parametrize("T,B,reverse,peep", [(23, 9, False, True), (8, 4, True, True), (3, 2, False, False), (1, 1, True, True),
                                              (41, 5, True, False), (100, 33, False, True)])
def test_lstm_fused16_vs_oracle(oracle, n, I, T, B, reverse, peep):
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(n + T + I)
    iW, sW, b, p = _params(rs, I, n, scale=2.0)
    if not peep:
        p = np.zeros_like(p)
        b = np.zeros_like(b)
    x = (rs.normal(size=(T, B, I)) * rs.choice([0.01, 1.0, 30.0], size=(T, B, 1))).astype(np.float32)   # rows of very different size
    ref = oracle.lstm(x, iW, sW, b, p, reverse=reverse)
    xw = torch.full((T, B, I + 8), np.nan, dtype=torch.float32, device="cuda")       # input and output as slices of wider tensors
    xw[:, :, :I] = dev(x)
    yw = torch.full((T, B, n + 16), np.nan, dtype=torch.float32, device="cuda")
    assert _layer(L, xw, I + 8, dev(iW), dev(sW), dev(b) if peep else None, dev(p) if peep else None, yw, n + 16, T, B, I, n,
                  reverse) == 0
    out = yw.cpu().numpy()
    assert np.isnan(out[:, :, n:]).all()
    err = np.abs(out[:, :, :n] - ref).max()
    assert err < 2e-5, err


@pytest.mark.parametrize("n,I", [(32, 12), (64, 12), (64, 64)])
def test_lstm_fused16_ragged(oracle, n, I):
    """Each chunk of a ragged batch equals the call on the chunk alone at its own length, reversed scans included; rows past a
    chunk's end stay untouched."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    T = 29
    rs = np.random.RandomState(n)
    lens = [29, 1, 20, 8, 28, 9, 2]
    B = len(lens)
    iW, sW, b, p = _params(rs, I, n, scale=2.0)
    x = np.zeros((T, B, I), dtype=np.float32)
    for bb, tb in enumerate(lens):
        x[:tb, bb] = rs.normal(size=(tb, I))
    ld = dev(np.asarray(lens, dtype=np.int32))
    for reverse in (False, True):
        y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
        assert _layer(L, dev(x), I, dev(iW), dev(sW), dev(b), dev(p), y, n, T, B, I, n, reverse, lens=ld) == 0
        out = y.cpu().numpy()
        for bb, tb in enumerate(lens):
            want = oracle.lstm(x[:tb, bb:bb + 1], iW, sW, b, p, reverse=reverse)
            np.testing.assert_allclose(out[:tb, bb:bb + 1], want, atol=2e-5, err_msg="chunk %d" % bb)
            assert np.isnan(out[tb:, bb]).all()


@pytest.mark.parametrize("I", [12, 64])
def test_lstm_fused16_large_weights_and_determinism(oracle, I):
    """|w| up to 6 with saturating gates; every launch must reproduce the first bit for bit (waves exchange the state and the
    projected inputs through LDS; the x rows arrive through requests the kernel counts itself)."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    n, T, B = 64, 61, 1021
    rs = np.random.RandomState(5)
    iW, sW, b, p = _params(rs, I, n, scale=2.0)
    sW[rs.randint(0, 4 * n, 60), rs.randint(0, n, 60)] = rs.choice([-6.0, 6.0, 4.5], size=60)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    xd, iWd, sWd, bd, pd = dev(x), dev(iW), dev(sW), dev(b), dev(p)
    first = None
    for rep in range(4):
        y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
        assert _layer(L, xd, I, iWd, sWd, bd, pd, y, n, T, B, I, n, True) == 0
        if first is None:
            first = y
        else:
            assert torch.equal(first, y)
    pick = [0, 3, 500, 1020]
    ref = oracle.lstm(x[:, pick], iW, sW, b, p, reverse=True)
    assert np.abs(first.cpu().numpy()[:, pick] - ref).max() < 5e-5


def test_lstm_fused16_is_what_the_layer_runs(oracle):
    """layers.Lstm takes this kernel for models/baseline_lstm.py's shapes and agrees with projection GEMM + scan."""
    torch = need_gpu()
    from sloika_amd import layers, profiler
    rs = np.random.RandomState(2)
    T, B, I, n = 50, 37, 12, 64
    net = layers.Lstm(I, n, init=lambda shape: rs.normal(size=shape).astype(np.float32), has_bias=True, has_peep=True)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    profiler.start()
    try:
        got = net.run(dev(x))
    finally:
        rec = profiler.stop()
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert "lstm_fused" in rec.summary() and "lstm_recurrent" not in rec.summary()
    saved, layers.LSTM_FUSED = layers.LSTM_FUSED, False
    try:
        two = net.run(dev(x)).cpu().numpy()
    finally:
        layers.LSTM_FUSED = saved
    assert np.abs(got - two).max() < 1e-5
    ref = oracle.lstm(x, net.iW.get_value(), net.sW.get_value(), net.b.get_value(), net.p.get_value(), reverse=False)
    assert np.abs(got - ref).max() < 2e-5


def test_lstm_fused16_unsupported_shapes_are_refused():
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    z = torch.zeros(65536, device="cuda")
    for n, I, ldx, act, gate in [(80, 12, 12, 1, 2), (24, 12, 12, 1, 2), (64, 68, 68, 1, 2), (64, 10, 12, 1, 2), (64, 12, 13, 1, 2),
                                 (64, 12, 12, 2, 2), (64, 12, 12, 1, 1)]:
        assert _layer(L, z, ldx, z, z, z, z, z, n, 1, 1, I, n, 0, act=act, gate=gate) == _lib.SLK_ERR_UNSUPPORTED
    assert _layer(L, z[1:], 12, z, z, z, z, z, 64, 1, 1, 12, 64, 0) == _lib.SLK_ERR_UNSUPPORTED          # x not 16-byte aligned
    assert L.slk_lstm_fused16_f32(None, 12, z.data_ptr(), z.data_ptr(), None, None, z.data_ptr(), 64, 1, 1, 12, 64, 0, 1, 2, None,
                                  stream()) == _lib.SLK_ERR_INVALID_ARG


def test_bilstm_directions_side_by_side(oracle):
    """The two directions of a birnn of fused Lstm layers run on two streams as long as two workgroups per CU hold them
    (B = 600: 2 x 150 workgroups on 256 CUs); the result is that of the layers one after the other."""
    torch = need_gpu()
    from sloika_amd import layers
    rs = np.random.RandomState(11)
    T, B, I, n = 40, 600, 12, 64
    init = lambda shape: rs.normal(size=shape).astype(np.float32)
    fwd = layers.Lstm(I, n, init=init, has_bias=True, has_peep=True)
    bwd = layers.Lstm(I, n, init=init, has_bias=True, has_peep=True)
    net = layers.birnn(fwd, bwd)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    xd = dev(x)
    assert net._side_streams(xd, B) is not None
    got = net.run(xd)
    torch.cuda.synchronize()
    again = net.run(xd)
    assert torch.equal(got, again)
    got = got.cpu().numpy()
    pick = [0, 1, 299, 598, 599]
    for layer, sl, reverse in ((fwd, slice(0, n), False), (bwd, slice(n, 2 * n), True)):
        ref = oracle.lstm(x[:, pick], layer.iW.get_value(), layer.sW.get_value(), layer.b.get_value(), layer.p.get_value(),
                          reverse=reverse)
        assert np.abs(got[:, pick, sl] - ref).max() < 2e-5
    saved, layers.LSTM_FUSED = layers.LSTM_FUSED, False
    try:
        assert net._side_streams(xd, B) is None          # 2 x 150 four-chunk workgroups of the scan kernel do not fit 256 CUs
        two = net.run(xd).cpu().numpy()
    finally:
        layers.LSTM_FUSED = saved
    assert np.abs(got - two).max() < 1e-5
