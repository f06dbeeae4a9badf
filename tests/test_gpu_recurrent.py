"""GPU parity: Gru (MFMA 4x4x1 kernel and portable kernel) and Lstm vs the oracle, through the C ABI.

Tolerance: 1e-4 absolute on layer outputs (BASELINE.json north_star); observed errors are ~1e-6.
"""
import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _gru_params(rs, I, n, bias=True, scale=1.0):
    iW = (rs.normal(size=(3 * n, I)) / np.sqrt(I + n)).astype(np.float32)
    sW = (scale * rs.normal(size=(2 * n, n)) / np.sqrt(2 * n)).astype(np.float32)
    sW2 = (scale * rs.normal(size=(n, n)) / np.sqrt(2 * n)).astype(np.float32)
    b = rs.normal(size=3 * n).astype(np.float32) if bias else None
    return iW, sW, sW2, b


@pytest.mark.parametrize("n", [16, 32, 48, 64, 80, 96, 112, 128, 144])
@pytest.mark.parametrize("reverse", [False, True])
def test_gru_mfma_all_sizes(oracle, n, reverse):
    torch = need_gpu()
    from sloika_amd import _lib
    rs = np.random.RandomState(n + reverse)
    T, B = 23, 9                                  # ragged last tile of 4 chunks
    iW, sW, sW2, b = _gru_params(rs, 12, n, scale=2.0)
    x = rs.normal(size=(T, B, 12)).astype(np.float32)
    vI = (x.astype(np.float64) @ iW.astype(np.float64).T + b).astype(np.float32)
    ref = oracle.gru(x, iW, sW, sW2, b, reverse=reverse)
    L = _lib.lib()
    vId, sWd, sW2d = dev(vI), dev(sW), dev(sW2)   # keep the device buffers alive across the calls
    for force_generic in (0, 1):
        y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
        rc = L.slk_gru_recurrent_f32_ex(vId.data_ptr(), sWd.data_ptr(), sW2d.data_ptr(), y.data_ptr(), n,
                                        T, B, n, int(reverse), 1, 2, force_generic, stream())
        assert rc == 0
        np.testing.assert_allclose(y.cpu().numpy(), ref, atol=TOL, err_msg="generic=%d" % force_generic)


@pytest.mark.parametrize("I,n,B,T", [(96, 96, 8, 60), (64, 64, 4, 50), (128, 112, 5, 40), (112, 144, 3, 30),
                                     (144, 112, 3, 30), (128, 110, 2, 20), (12, 4, 2, 25), (7, 5, 1, 9)])
def test_gru_layer_vs_oracle(oracle, I, n, B, T):
    need_gpu()
    from sloika_amd import layers
    rs = np.random.RandomState(I + n)
    iW, sW, sW2, b = _gru_params(rs, I, n, scale=2.0)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    g = layers.Gru(I, n, has_bias=True)
    g.set_params({"iW": iW.reshape(3, n, I), "sW": sW.reshape(2, n, n), "sW2": sW2, "b": b.reshape(3, n)})
    for net in (g, layers.Reverse(g)):
        y = net.compile()(x)
        ref = oracle.run_network(net.spec(), x)
        np.testing.assert_allclose(y, ref, atol=TOL)


def test_gru_other_activations_and_no_bias(oracle):
    need_gpu()
    from sloika_amd import layers, activation
    rs = np.random.RandomState(9)
    I, n, T, B = 10, 32, 15, 4
    iW, sW, sW2, _ = _gru_params(rs, I, n, bias=False)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    g = layers.Gru(I, n, has_bias=False, fun=activation.retu, gatefun=activation.sigmoid_pm)
    g.set_params({"iW": iW.reshape(3, n, I), "sW": sW.reshape(2, n, n), "sW2": sW2})
    np.testing.assert_allclose(g.compile()(x), oracle.run_network(g.spec(), x), atol=TOL)


def test_gru_strided_output_slice(oracle):
    """birnn writes both directions into one concatenated tensor (layers.py:1486-1487, 1622-1629)."""
    need_gpu()
    from sloika_amd import layers
    rs = np.random.RandomState(4)
    I, n, T, B = 16, 64, 30, 6
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    gs = []
    for _ in range(2):
        iW, sW, sW2, b = _gru_params(rs, I, n, scale=1.5)
        g = layers.Gru(I, n, has_bias=True)
        g.set_params({"iW": iW.reshape(3, n, I), "sW": sW.reshape(2, n, n), "sW2": sW2, "b": b.reshape(3, n)})
        gs.append(g)
    net = layers.birnn(gs[0], gs[1])
    y = net.compile()(x)
    assert y.shape == (T, B, 2 * n)
    np.testing.assert_allclose(y, oracle.run_network(net.spec(), x), atol=TOL)


@pytest.mark.parametrize("I,n,bias,peep,T,B", [(12, 64, True, True, 20, 3), (5, 16, False, False, 20, 3),
                                               (64, 96, True, False, 20, 3), (3, 7, True, True, 20, 3),
                                               (12, 64, True, True, 61, 9), (8, 32, True, True, 33, 5),
                                               (9, 48, False, True, 1, 1), (64, 64, True, False, 2, 4)])
@pytest.mark.parametrize("reverse", [False, True])
def test_lstm_vs_oracle(oracle, I, n, bias, peep, T, B, reverse):
    """Multiples of 16 up to 128 take the fp16-split scan (csrc/lstm_scan16.hip), the others the portable kernel."""
    need_gpu()
    from sloika_amd import layers
    rs = np.random.RandomState(I * 10 + n)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    l = layers.Lstm(I, n, has_bias=bias, has_peep=peep)
    l.iW.set_value((rs.normal(size=(4 * n, I)) / np.sqrt(I + n)).astype(np.float32))
    l.sW.set_value((rs.normal(size=(4 * n, n)) / np.sqrt(2 * n)).astype(np.float32))
    if bias:
        l.b.set_value(rs.normal(size=4 * n).astype(np.float32))
    if peep:
        l.p.set_value((rs.normal(size=(3, n)) / np.sqrt(n)).astype(np.float32))
    net = layers.Reverse(l) if reverse else l
    np.testing.assert_allclose(net.compile()(x), oracle.run_network(net.spec(), x), atol=TOL)


@pytest.mark.parametrize("n", [64, 32, 7])
def test_lstm_ragged_reverse_vs_oracle(oracle, n):
    """Ragged batch through a bidirectional Lstm: chunk b, padded to T, must see exactly what it sees alone at its own
    length (the reversed scan starts at the chunk's own last step); steps past the end are not compared."""
    torch = need_gpu()
    from sloika_amd import layers
    rs = np.random.RandomState(n)
    I, T = 6, 37
    lens = [37, 1, 20, 8, 36, 9, 2]
    x = np.zeros((T, len(lens), I), dtype=np.float32)
    for b, tb in enumerate(lens):
        x[:tb, b] = rs.normal(size=(tb, I))
    ls = []
    for _ in range(2):
        l = layers.Lstm(I, n)
        l.iW.set_value((rs.normal(size=(4 * n, I)) / np.sqrt(I + n)).astype(np.float32))
        l.sW.set_value((rs.normal(size=(4 * n, n)) / np.sqrt(2 * n)).astype(np.float32))
        l.b.set_value(rs.normal(size=4 * n).astype(np.float32))
        l.p.set_value((rs.normal(size=(3, n)) / np.sqrt(n)).astype(np.float32))
        ls.append(l)
    net = layers.birnn(ls[0], ls[1])
    with layers.ragged(lens):
        y = net.run(torch.from_numpy(x).cuda()).cpu().numpy()
    for b, tb in enumerate(lens):
        want = oracle.run_network(net.spec(), x[:tb, b:b + 1])
        np.testing.assert_allclose(y[:tb, b:b + 1], want, atol=TOL, err_msg="chunk %d" % b)


def test_layer_input_validation():
    torch = need_gpu()
    from sloika_amd import layers
    g = layers.Gru(4, 16)
    with pytest.raises(ValueError):
        g.run(torch.zeros((5, 2, 3), device="cuda"))           # wrong feature count
    with pytest.raises(ValueError):
        g.run(torch.zeros((5, 2, 4)))                          # not on the device


@pytest.mark.parametrize("I,n", [(96, 96), (64, 64), (32, 96), (64, 96), (16, 16), (48, 32), (16, 64)])
@pytest.mark.parametrize("T,B,reverse", [(23, 9, False), (8, 4, True), (3, 2, False), (1, 1, True), (41, 5, True)])
def test_gru_layer_entry(oracle, I, n, T, B, reverse):
    """slk_gru_f32 -- the whole layer in one persistent kernel where an instantiation exists (csrc/gru_bar16.hip), projection
    GEMM + scan otherwise (16 -> 16) -- vs the oracle."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    ws = torch.empty(L.slk_gru_workspace_bytes(T, B, n), dtype=torch.uint8, device="cuda")
    rs = np.random.RandomState(I + n + T)
    iW, sW, sW2, b = _gru_params(rs, I, n, scale=2.0)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    ref = oracle.gru(x, iW, sW, sW2, b, reverse=reverse)
    xd, iWd, sWd, sW2d, bd = dev(x), dev(iW), dev(sW), dev(sW2), dev(b)
    y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
    rc = L.slk_gru_f32(xd.data_ptr(), I, iWd.data_ptr(), sWd.data_ptr(), sW2d.data_ptr(), bd.data_ptr(),
                       y.data_ptr(), n, T, B, I, n, int(reverse), 1, 2, ws.data_ptr(), ws.numel(), stream())
    assert rc == 0
    np.testing.assert_allclose(y.cpu().numpy(), ref, atol=TOL)
    # no bias + strided input/output rows (slices of wider tensors)
    xw = torch.zeros((T, B, I + 16), device="cuda")
    xw[:, :, 8:8 + I] = xd
    yw = torch.full((T, B, n + 8), -5.0, device="cuda")
    rc = L.slk_gru_f32(xw.data_ptr() + 8 * 4, I + 16, iWd.data_ptr(), sWd.data_ptr(), sW2d.data_ptr(), None,
                       yw.data_ptr() + 4 * 4, n + 8, T, B, I, n, int(reverse), 1, 2, ws.data_ptr(), ws.numel(), stream())
    assert rc == 0
    ref_nb = oracle.gru(x, iW, sW, sW2, None, reverse=reverse)
    out = yw.cpu().numpy()
    np.testing.assert_allclose(out[:, :, 4:4 + n], ref_nb, atol=TOL)
    assert (out[:, :, :4] == -5.0).all() and (out[:, :, 4 + n:] == -5.0).all()


@pytest.mark.parametrize("I,n", [(96, 96), (128, 96), (112, 112), (64, 64), (112, 144), (128, 128)])
def test_gru_kernels_agree_under_load(I, n):
    """Race screen: many tiles, several hundred steps (dozens of LDS ring turnovers), every CU busy, repeated.  The
    portable kernel, the MFMA recurrence and (where instantiated) the fused layer kernel must agree; any slot reused
    too early shows up as a large error in some chunk."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    T, B = 333, 1021
    g = torch.Generator(device="cuda").manual_seed(I * n)
    x = torch.randn((T, B, I), device="cuda", generator=g)
    iW = torch.randn((3 * n, I), device="cuda", generator=g) / np.sqrt(I + n)
    sW = 2.0 * torch.randn((2 * n, n), device="cuda", generator=g) / np.sqrt(2 * n)
    sW2 = 2.0 * torch.randn((n, n), device="cuda", generator=g) / np.sqrt(2 * n)
    b = torch.randn(3 * n, device="cuda", generator=g)
    vI = torch.empty((T, B, 3 * n), device="cuda")
    assert L.slk_gemm_bias_act_f32(x.data_ptr(), I, iW.data_ptr(), b.data_ptr(), vI.data_ptr(), 3 * n, T * B, I, 3 * n, 0,
                                   stream()) == 0
    for reverse in (0, 1):
        ref = torch.empty((T, B, n), device="cuda")
        assert L.slk_gru_recurrent_f32_ex(vI.data_ptr(), sW.data_ptr(), sW2.data_ptr(), ref.data_ptr(), n, T, B, n, reverse,
                                          1, 2, 1, stream()) == 0
        for rep in range(3):
            y = torch.full((T, B, n), float("nan"), device="cuda")
            assert L.slk_gru_recurrent_f32_ex(vI.data_ptr(), sW.data_ptr(), sW2.data_ptr(), y.data_ptr(), n, T, B, n,
                                              reverse, 1, 2, 0, stream()) == 0
            err = (y - ref).abs().max().item()
            assert err < TOL, "recurrent kernel, reverse=%d rep=%d: %g" % (reverse, rep, err)
            y.fill_(float("nan"))
            rc = L.slk_gru_bar16_f32(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), b.data_ptr(),
                                     y.data_ptr(), n, T, B, I, n, reverse, 1, 2, None, None, stream())
            assert rc in (0, _lib.SLK_ERR_UNSUPPORTED)
            if rc == 0:
                err = (y - ref).abs().max().item()
                assert err < TOL, "fused kernel, reverse=%d rep=%d: %g" % (reverse, rep, err)


def test_gru_layer_kernel_unsupported_shapes(oracle):
    need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    z = dev(np.zeros((4, 4), dtype=np.float32))
    # sizes and activations with no instantiation report UNSUPPORTED (slk_gru_f32 then takes the two-kernel path)
    assert L.slk_gru_bar16_f32(z.data_ptr(), 7, z.data_ptr(), z.data_ptr(), z.data_ptr(), None, z.data_ptr(), 5, 1, 1, 7, 5,
                               0, 1, 2, None, None, stream()) == _lib.SLK_ERR_UNSUPPORTED
    assert L.slk_gru_bar16_f32(z.data_ptr(), 96, z.data_ptr(), z.data_ptr(), z.data_ptr(), None, z.data_ptr(), 96, 1, 1, 96,
                               96, 0, 3, 2, None, None, stream()) == _lib.SLK_ERR_UNSUPPORTED
