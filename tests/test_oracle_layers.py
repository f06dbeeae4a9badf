"""Oracle layer maths: restates the reference's numpy known-answer tests (test/unit/test_layers.py) and
cross-checks the C oracle (float32) against the independent float64 numpy restatement.

Gru / Lstm / Convolution are pinned to the reference's own layer code in tests/test_oracle_reference_layers.py.
"""
import numpy as np
import pytest

from oracle import oracle_np

_NSTEP, _NFEATURES, _SIZE, _NBATCH = 25, 3, 64, 2       # test_layers.py:47-56


@pytest.fixture(scope="module")
def ann():
    np.random.seed(0xdeadbeef)
    W = np.random.normal(size=(_SIZE, _NFEATURES)).astype(np.float32)
    b = np.random.normal(size=_SIZE).astype(np.float32)
    x = np.random.normal(size=(_NSTEP, _NBATCH, _NFEATURES)).astype(np.float32)
    return W, b, x, x.dot(W.transpose()) + b


def test_feedforward_linear_and_tanh(oracle, ann):                  # test_layers.py:58-69
    W, b, x, res = ann
    np.testing.assert_almost_equal(oracle.feedforward(x, W, b, "linear"), res, decimal=5)
    np.testing.assert_almost_equal(oracle.feedforward(x, W, b, "tanh"), np.tanh(res), decimal=5)


def test_serial(oracle, ann):                                         # test_layers.py:82-94
    W, b, x, res = ann
    W2 = np.random.RandomState(1).normal(size=(_SIZE, _SIZE)).astype(np.float32)
    spec = {"type": "serial", "sublayers": [
        {"type": "feed-forward", "W": W, "b": b, "activation": "linear"},
        {"type": "feed-forward", "W": W2, "b": None, "activation": "linear"}]}
    np.testing.assert_almost_equal(oracle.run_network(spec, x), res.dot(W2.transpose()), decimal=4)


def test_reverse_and_birnn_of_timelocal_layer(oracle, ann):          # test_layers.py:96-116
    W, b, x, res = ann
    ff = {"type": "feed-forward", "W": W, "b": b, "activation": "tanh"}
    r1 = oracle.run_network(ff, x)
    r2 = oracle.run_network({"type": "reverse", "sublayer": ff}, x)
    np.testing.assert_almost_equal(r1, r2)
    bi = oracle.run_network({"type": "parallel", "sublayers": [ff, {"type": "reverse", "sublayer": ff}]}, x)
    np.testing.assert_almost_equal(bi[:, :, :_SIZE], bi[:, :, _SIZE:])


def test_softmax_rows_sum_to_one(oracle, ann):                       # test_layers.py:118-125
    W, b, x, _ = ann
    res = oracle.softmax(x, W, b)
    assert np.allclose(res.sum(axis=2), 1.0)
    np.testing.assert_allclose(res, oracle_np.softmax(x, W, b), atol=1e-6)


def test_window_layout(oracle, ann):                                 # test_layers.py:246-266
    _, _, x, _ = ann
    w = 3
    res_full = oracle.window(x, w)
    assert np.array_equal(res_full, oracle_np.window(x, w))
    res = res_full[w // 2: -(w // 2)]
    for j in range(_NBATCH):
        for i in range(w - 1):
            np.testing.assert_almost_equal(res[:, j, i * w:(i + 1) * w], x[i:1 + i - w, j])
        np.testing.assert_almost_equal(res[:, j, w * (w - 1):], x[w - 1:, j])
        np.testing.assert_almost_equal(x[:w, j].ravel(), res[0, j].transpose().ravel())
        np.testing.assert_almost_equal(x[-w:, j].ravel(), res[-1, j].transpose().ravel())


def test_activations_match_numpy(oracle):
    x = np.linspace(-6, 6, 97).astype(np.float32)
    for name in oracle.ACTIVATIONS:
        np.testing.assert_allclose(oracle.activation(name, x), oracle_np.ACT[name](x.astype(np.float64)),
                                   rtol=2e-6, atol=2e-6, err_msg=name)


@pytest.mark.parametrize("T,B,Cin,Cout,w,s,pad,act", [
    (100, 3, 1, 8, 11, 5, (5, 5), "elu"),
    (101, 2, 1, 6, 11, 2, (5, 5), "tanh"),
    (40, 2, 12, 32, 11, 5, (5, 5), "tanh"),          # test_layers.py Convolution(12,32,11,5)
    (30, 2, 3, 4, 4, 1, (1, 2), "linear"),           # even window, 'same' padding (conv.py:52-53)
    (30, 1, 2, 3, 5, 3, (0, 0), "relu"),             # 'valid'
])
def test_conv1d_c_vs_numpy(oracle, T, B, Cin, Cout, w, s, pad, act):
    rs = np.random.RandomState(T + Cin)
    x = rs.normal(size=(T, B, Cin)).astype(np.float32)
    W = rs.normal(size=(Cout, Cin, w)).astype(np.float32) * 0.3
    b = rs.normal(size=Cout).astype(np.float32)
    y = oracle.conv1d(x, W, b, s, pad, act)
    y64 = oracle_np.conv1d(x, W, b, s, pad, act)
    assert y.shape == y64.shape
    assert y.shape[0] == (T + pad[0] + pad[1] - w) // s + 1
    np.testing.assert_allclose(y, y64, atol=2e-5)


@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("I,n,bias", [(12, 4, True), (7, 16, False), (96, 96, True)])
def test_gru_c_vs_numpy(oracle, I, n, bias, reverse):
    rs = np.random.RandomState(I * 100 + n)
    T, B = 20, 3
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    iW = (rs.normal(size=(3 * n, I)) / np.sqrt(I + n)).astype(np.float32)
    sW = (rs.normal(size=(2 * n, n)) / np.sqrt(2 * n)).astype(np.float32) * 2
    sW2 = (rs.normal(size=(n, n)) / np.sqrt(2 * n)).astype(np.float32) * 2
    b = rs.normal(size=3 * n).astype(np.float32) if bias else None
    y = oracle.gru(x, iW, sW, sW2, b, reverse=reverse)
    y64 = oracle_np.gru(x, iW, sW, sW2, b, reverse=reverse)
    np.testing.assert_allclose(y, y64, atol=2e-5)
    # Reverse(layer).run(x) == layer.run(x[::-1])[::-1]      (layers.py:1449-1450)
    if reverse:
        np.testing.assert_array_equal(y, oracle.gru(x[::-1], iW, sW, sW2, b)[::-1])


def test_gru_zero_weights_is_zero_state(oracle):
    # h0 = 0 and hbar = tanh(0) = 0 -> output stays 0 whatever z is (layers.py:1019-1020)
    x = np.ones((5, 2, 3), dtype=np.float32)
    z = np.zeros
    y = oracle.gru(x, z((12, 3)), z((8, 4)), z((4, 4)), None)
    assert np.array_equal(y, np.zeros((5, 2, 4), dtype=np.float32))


@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("I,n,bias,peep", [(12, 4, True, True), (5, 16, False, False), (64, 64, True, False)])
def test_lstm_c_vs_numpy(oracle, I, n, bias, peep, reverse):
    rs = np.random.RandomState(I * 10 + n)
    T, B = 15, 2
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    iW = (rs.normal(size=(4 * n, I)) / np.sqrt(I + n)).astype(np.float32)
    sW = (rs.normal(size=(4 * n, n)) / np.sqrt(2 * n)).astype(np.float32)
    b = rs.normal(size=4 * n).astype(np.float32) if bias else None
    p = (rs.normal(size=(3, n)) / np.sqrt(n)).astype(np.float32) if peep else None
    y = oracle.lstm(x, iW, sW, b, p, reverse=reverse)
    y64 = oracle_np.lstm(x, iW, sW, b, p, reverse=reverse)
    np.testing.assert_allclose(y, y64, atol=2e-5)


def test_network_graph_c_vs_numpy(oracle):
    """A small bidirectional stack shaped like models/baseline_raw_gru.py:21-37."""
    rs = np.random.RandomState(7)
    size, nst = 8, 65

    def gru_spec(i, n):
        return {"type": "GRU", "iW": rs.normal(size=(3 * n, i)).astype(np.float32) * 0.3,
                "sW": rs.normal(size=(2 * n, n)).astype(np.float32) * 0.3,
                "sW2": rs.normal(size=(n, n)).astype(np.float32) * 0.3,
                "b": rs.normal(size=3 * n).astype(np.float32) * 0.1, "activation": "tanh", "gate": "sigmoid"}

    spec = {"type": "serial", "sublayers": [
        {"type": "convolution", "W": rs.normal(size=(size, 1, 11)).astype(np.float32) * 0.3,
         "b": rs.normal(size=size).astype(np.float32) * 0.1, "stride": 2, "padding": (5, 5), "activation": "tanh"},
        {"type": "parallel", "sublayers": [gru_spec(size, size), {"type": "reverse", "sublayer": gru_spec(size, size)}]},
        {"type": "feed-forward", "W": rs.normal(size=(size, 2 * size)).astype(np.float32) * 0.3,
         "b": rs.normal(size=size).astype(np.float32) * 0.1, "activation": "tanh"},
        {"type": "softmax", "W": rs.normal(size=(nst, size)).astype(np.float32) * 0.3,
         "b": rs.normal(size=nst).astype(np.float32) * 0.1}]}
    x = rs.normal(size=(60, 3, 1)).astype(np.float32)
    y = oracle.run_network(spec, x)
    y64 = oracle_np.run_network(spec, x)
    assert y.shape == (30, 3, nst)
    np.testing.assert_allclose(y, y64, atol=1e-5)
