"""GPU parity: fp32-MFMA projections (FeedForward, Softmax) through the C ABI vs the oracle."""
import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu


def test_feedforward_reference_kats(oracle):
    # test/unit/test_layers.py:47-69, 82-94
    need_gpu()
    from sloika_amd import layers, activation
    np.random.seed(0xdeadbeef)
    W = np.random.normal(size=(64, 3)).astype(np.float32)
    b = np.random.normal(size=64).astype(np.float32)
    x = np.random.normal(size=(25, 2, 3)).astype(np.float32)
    res = x.dot(W.transpose()) + b
    net = layers.FeedForward(3, 64, has_bias=True, fun=activation.linear)
    net.set_params({'W': W, 'b': b})
    np.testing.assert_almost_equal(net.compile()(x), res, decimal=5)
    net = layers.FeedForward(3, 64, has_bias=True)
    net.set_params({'W': W, 'b': b})
    np.testing.assert_almost_equal(net.compile()(x), np.tanh(res), decimal=5)
    W2 = np.random.normal(size=(64, 64)).astype(np.float32)
    l1 = layers.FeedForward(3, 64, has_bias=True, fun=activation.linear)
    l1.set_params({'W': W, 'b': b})
    l2 = layers.FeedForward(64, 64, fun=activation.linear)
    l2.set_params({'W': W2})
    np.testing.assert_almost_equal(layers.Serial([l1, l2]).compile()(x), res.dot(W2.transpose()), decimal=4)
    # Parallel / birnn of a time-local layer: both halves equal (test_layers.py:71-80, 107-116)
    l3 = layers.FeedForward(3, 64, has_bias=True)
    l3.set_params({'W': W, 'b': b})
    l4 = layers.FeedForward(3, 64, has_bias=True)
    l4.set_params({'W': W, 'b': b})
    r = layers.Parallel([l3, l4]).compile()(x)
    np.testing.assert_almost_equal(r[:, :, :64], r[:, :, 64:])
    r = layers.birnn(l3, l4).compile()(x)
    np.testing.assert_almost_equal(r[:, :, :64], r[:, :, 64:])
    np.testing.assert_almost_equal(layers.Reverse(l3).compile()(x), l3.compile()(x))
    # softmax rows sum to one (test_layers.py:118-125)
    sm = layers.Softmax(3, 64, has_bias=True)
    sm.set_params({'W': W, 'b': b})
    assert np.allclose(sm.compile()(x).sum(axis=2), 1.0)


@pytest.mark.parametrize("M,K,N,act", [
    (1, 1, 1, "linear"), (50, 3, 64, "tanh"), (129, 7, 33, "linear"), (300, 96, 288, "linear"),
    (257, 64, 192, "linear"), (1000, 128, 64, "tanh"), (200, 192, 128, "tanh"), (77, 96, 1025, "linear"),
    (128, 33, 97, "sigmoid"), (4096, 96, 288, "linear"),
])
def test_gemm_bias_act_vs_float64(M, K, N, act):
    torch = need_gpu()
    from sloika_amd import _lib
    from oracle import oracle_np
    rs = np.random.RandomState(M + K + N)
    x = rs.normal(size=(M, K)).astype(np.float32)
    W = (rs.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rs.normal(size=N).astype(np.float32)
    y = torch.empty((M, N), dtype=torch.float32, device="cuda")
    xd, Wd, bd = dev(x), dev(W), dev(b)            # keep the device buffers alive across the call
    rc = _lib.lib().slk_gemm_bias_act_f32(xd.data_ptr(), K, Wd.data_ptr(), bd.data_ptr(), y.data_ptr(), N,
                                          M, K, N, {"linear": 0, "tanh": 1, "sigmoid": 2}[act], stream())
    assert rc == 0
    ref = oracle_np.ACT[act](x.astype(np.float64) @ W.astype(np.float64).T + b)
    np.testing.assert_allclose(y.cpu().numpy(), ref, atol=5e-6, rtol=1e-5)


def test_gemm_strided_rows_and_no_bias():
    torch = need_gpu()
    from sloika_amd import _lib
    rs = np.random.RandomState(3)
    M, K, N = 70, 40, 50
    xbig = rs.normal(size=(M, K + 24)).astype(np.float32)
    W = rs.normal(size=(N, K)).astype(np.float32)
    xd, Wd = dev(xbig), dev(W)
    ybig = torch.full((M, N + 14), -7.0, dtype=torch.float32, device="cuda")
    # read columns [8, 8+K) of x, write columns [6, 6+N) of y
    rc = _lib.lib().slk_gemm_bias_act_f32(xd.data_ptr() + 8 * 4, K + 24, Wd.data_ptr(), None,
                                          ybig.data_ptr() + 6 * 4, N + 14, M, K, N, 0, stream())
    assert rc == 0
    out = ybig.cpu().numpy()
    ref = xbig[:, 8:8 + K].astype(np.float64) @ W.astype(np.float64).T
    np.testing.assert_allclose(out[:, 6:6 + N], ref, atol=2e-5)
    assert np.all(out[:, :6] == -7.0) and np.all(out[:, 6 + N:] == -7.0)


def test_gemm_argument_errors():
    need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    x = dev(np.zeros((4, 4), dtype=np.float32))
    assert L.slk_gemm_bias_act_f32(None, 4, x.data_ptr(), None, x.data_ptr(), 4, 4, 4, 4, 0, stream()) == _lib.SLK_ERR_INVALID_ARG
    assert L.slk_gemm_bias_act_f32(x.data_ptr(), 2, x.data_ptr(), None, x.data_ptr(), 4, 4, 4, 4, 0, stream()) == _lib.SLK_ERR_INVALID_ARG
    assert L.slk_gemm_bias_act_f32(x.data_ptr(), 4, x.data_ptr(), None, x.data_ptr(), 4, 4, 4, 4, 99, stream()) == _lib.SLK_ERR_INVALID_ARG
    with pytest.raises(ValueError):
        _lib.check(_lib.SLK_ERR_INVALID_ARG, "x")


@pytest.mark.parametrize("T,B,I,N", [(25, 2, 3, 64), (40, 3, 96, 1025), (10, 2, 64, 126), (6, 1, 8, 5), (3, 2, 16, 1400)])
def test_softmax_vs_oracle(oracle, T, B, I, N):
    need_gpu()
    from sloika_amd import layers
    rs = np.random.RandomState(N)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    sm = layers.Softmax(I, N, has_bias=True)
    sm.set_params({"W": (rs.normal(size=(N, I)) * 0.5).astype(np.float32), "b": rs.normal(size=N).astype(np.float32)})
    y = sm.compile()(x)
    ref = oracle.run_network(sm.spec(), x)
    np.testing.assert_allclose(y, ref, atol=2e-6, rtol=1e-5)
    assert np.allclose(y.sum(axis=2), 1.0, atol=1e-5)


@pytest.mark.parametrize("M,K,N,ld", [(1, 1, 1, 1), (130, 96, 1025, 1056), (257, 64, 126, 126), (64, 33, 97, 100),
                                      (300, 128, 200, 224), (129, 112, 1025, 1025), (5, 7, 3, 8)])
@pytest.mark.parametrize("kernel", ["f32", "f16x3"])
def test_linear_rowstats_kernels_vs_float64(M, K, N, ld, kernel):
    """x-stationary projection + online softmax statistics: plain fp32 MFMA and the 3-term fp16 split.  Both must sit
    within a few float32 ulps of a float64 evaluation (trained-weight magnitudes: |w| up to 6)."""
    torch = need_gpu()
    from sloika_amd import _lib
    rs = np.random.RandomState(M + K + N)
    x = np.tanh(rs.normal(size=(M, K))).astype(np.float32)
    W = (rs.normal(size=(N, K)) * 1.5).astype(np.float32)
    W[0, :] *= 4.0
    b = rs.normal(size=N).astype(np.float32)
    xd, Wd, bd = dev(x), dev(W), dev(b)
    y = torch.full((M, ld), np.nan, dtype=torch.float32, device="cuda")
    stats = torch.empty((M, 2), dtype=torch.float32, device="cuda")
    L = _lib.lib()
    if kernel == "f32":
        rc = L.slk_linear_rowstats_f32(xd.data_ptr(), K, Wd.data_ptr(), bd.data_ptr(), y.data_ptr(), ld, M, K, N,
                                       stats.data_ptr(), stream())
    else:
        KP = (K + 15) // 16 * 16
        hi = torch.empty((N, KP), dtype=torch.float16, device="cuda")
        lo = torch.empty((N, KP), dtype=torch.float16, device="cuda")
        inv = torch.empty((N,), dtype=torch.float32, device="cuda")
        assert L.slk_split_f16x2_f32(Wd.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), stream()) == 0
        # the two halves times the inverse row scale reproduce W to ~2^-22 of each row's largest weight
        back = (hi[:, :K].float() + lo[:, :K].float()) * inv[:, None]
        assert ((back - Wd).abs() <= 3e-7 * Wd.abs().amax(dim=1, keepdim=True)).all()
        assert (hi.float().abs().amax(dim=1) < 2.001).all() and (hi.float().abs().amax(dim=1) >= 1.0).all()
        rc = L.slk_linear_rowstats_f16x3(xd.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), bd.data_ptr(),
                                         y.data_ptr(), ld, M, K, N, stats.data_ptr(), stream())
    assert rc == 0
    out = y.cpu().numpy()
    ref = x.astype(np.float64) @ W.astype(np.float64).T + b
    scale = np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64).T + np.abs(b)
    assert (np.abs(out[:, :N] - ref) <= 6e-7 * scale + 1e-6).all()
    if ld > N:
        assert np.isnan(out[:, N:]).all()                 # padding columns are never written
    m = out[:, :N].max(axis=1)
    np.testing.assert_array_equal(stats[:, 0].cpu().numpy(), m)
    s = np.exp(out[:, :N].astype(np.float64) - m[:, None]).sum(axis=1)
    np.testing.assert_allclose(1.0 / stats[:, 1].cpu().numpy().astype(np.float64), s, rtol=3e-6)
    # no-statistics form writes the same logits
    y2 = torch.empty((M, ld), dtype=torch.float32, device="cuda")
    if kernel == "f32":
        rc = L.slk_linear_rowstats_f32(xd.data_ptr(), K, Wd.data_ptr(), bd.data_ptr(), y2.data_ptr(), ld, M, K, N, None, stream())
    else:
        rc = L.slk_linear_rowstats_f16x3(xd.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), bd.data_ptr(),
                                         y2.data_ptr(), ld, M, K, N, None, stream())
    assert rc == 0
    assert torch.equal(y2[:, :N], y[:, :N])


def test_linear_rowstats_unsupported_k():
    need_gpu()
    from sloika_amd import _lib
    z = dev(np.zeros((4, 200), dtype=np.float32))
    assert _lib.lib().slk_linear_rowstats_f32(z.data_ptr(), 200, z.data_ptr(), None, z.data_ptr(), 200, 4, 200, 4, None,
                                              stream()) == _lib.SLK_ERR_UNSUPPORTED


@pytest.mark.parametrize("K,N,act", [(192, 128, "tanh"), (128, 64, "tanh"), (8, 4, "tanh"), (96, 1025, "linear"),
                                     (176, 70, "sigmoid"), (33, 65, "relu"), (144, 336, "elu")])
def test_gemm_bias_act_f16x3_vs_float64(K, N, act):
    """FeedForward on the fp16 pipe (3-term split): every supported activation, K up to 192, ragged N and M, strided rows."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(K + N)
    M = 1000 + K
    x = rs.normal(size=(M, K)).astype(np.float32)
    W = (rs.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rs.normal(size=N).astype(np.float32)
    z = x.astype(np.float64) @ W.astype(np.float64).T + b
    ref = {"tanh": np.tanh, "linear": lambda v: v, "sigmoid": lambda v: 1.0 / (1.0 + np.exp(-v)),
           "relu": lambda v: np.maximum(v, 0.0), "elu": lambda v: np.where(v > 0, v, np.expm1(v))}[act](z)
    aid = {"linear": 0, "tanh": 1, "sigmoid": 2, "elu": 3, "relu": 4}[act]
    xd, Wd, bd = dev(x), dev(W), dev(b)
    kp = (K + 15) // 16 * 16
    hi = torch.empty((N, kp), dtype=torch.float16, device="cuda")
    lo = torch.empty((N, kp), dtype=torch.float16, device="cuda")
    inv = torch.empty((N,), dtype=torch.float32, device="cuda")
    assert L.slk_split_f16x2_f32(Wd.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), stream()) == 0
    ld = N + 5
    y = torch.full((M, ld), -7.0, dtype=torch.float32, device="cuda")
    rc = L.slk_gemm_bias_act_f16x3(xd.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), bd.data_ptr(), y.data_ptr(),
                                   ld, M, K, N, aid, stream())
    assert rc == 0
    out = y.cpu().numpy()
    np.testing.assert_allclose(out[:, :N], ref, atol=2e-5)
    assert (out[:, N:] == -7.0).all()
    # unsupported activation -> caller falls back to the fp32 kernel
    assert L.slk_gemm_bias_act_f16x3(xd.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), bd.data_ptr(), y.data_ptr(),
                                     ld, M, K, N, 7, stream()) == _lib.SLK_ERR_UNSUPPORTED


@pytest.mark.parametrize("xmag,wmag", [(1e-7, 1.0), (1e-3, 1e3), (1e3, 1e-3), (1e5, 1e-5), (1e5, 1.0), (3e7, 1e-6), (1.0, 1e6)])
def test_f16x3_operands_of_any_magnitude(xmag, wmag):
    """fp16 alone overflows at 65504 and has no lo half below 6e-5.  Rows of x and of W are scaled by powers of two before
    their split, so the 3-term product stays float32-grade for any finite magnitudes -- also when rows of very different
    size share a tile and when one row holds an outlier.  Checked against float64, relative to sum |x||w|."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    M, K, N = 300, 96, 1025
    rs = np.random.RandomState(17)
    x = (rs.normal(size=(M, K)) * xmag).astype(np.float32)
    x[5] *= 1e-4; x[6] *= 1e3; x[7, 11] *= 300.0; x[8] = 0.0
    W = (rs.normal(size=(N, K)) * wmag).astype(np.float32)
    W[3] *= 1e-5; W[4] *= 1e4; W[9, 2] *= 100.0; W[10] = 0.0
    b = (rs.normal(size=N) * xmag * wmag).astype(np.float32)
    xd, Wd, bd = dev(x), dev(W), dev(b)
    hi = torch.empty((N, K), dtype=torch.float16, device="cuda")
    lo = torch.empty((N, K), dtype=torch.float16, device="cuda")
    inv = torch.empty((N,), dtype=torch.float32, device="cuda")
    assert L.slk_split_f16x2_f32(Wd.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), stream()) == 0
    y = torch.full((M, N), np.nan, dtype=torch.float32, device="cuda")
    stats = torch.empty((M, 2), dtype=torch.float32, device="cuda")
    assert L.slk_linear_rowstats_f16x3(xd.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), bd.data_ptr(),
                                       y.data_ptr(), N, M, K, N, stats.data_ptr(), stream()) == 0
    out = y.cpu().numpy().astype(np.float64)
    assert np.isfinite(out).all()
    ref = x.astype(np.float64) @ W.astype(np.float64).T + b
    # error bound of a float32-grade product: a few 2^-24 of (row max |x|) * (row max |w|) * K, plus the terms' own sum
    bound = 1e-6 * (np.abs(x).max(axis=1, keepdims=True).astype(np.float64) * np.abs(W).max(axis=1)[None, :] * np.sqrt(K)
                    + np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64).T + np.abs(b))
    assert (np.abs(out - ref) <= bound + 1e-30).all(), float((np.abs(out - ref) / (bound + 1e-300)).max())
