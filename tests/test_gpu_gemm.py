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
