"""The training-step oracle (oracle/oracle_train.py) against finite differences of the
already-pinned forward oracle, and its ADAMski restatement against the closed form of the first steps."""
import numpy as np
import pytest

from oracle import oracle_train as ot


def _net(rs, nfeat=1, n=6, nstate=9, conv_act="elu", stride=2, winlen=5, bias=True):
    def r(*shape):
        return rs.normal(size=shape) * 0.5
    pad = ((winlen - 1) // 2, winlen // 2)
    gru = lambda i: {"type": "GRU", "iW": r(3 * n, i), "sW": r(2 * n, n), "sW2": r(n, n), "b": r(3 * n) if bias else None,
                     "activation": "tanh", "gate": "sigmoid"}
    return {"type": "serial", "sublayers": [
        {"type": "convolution", "W": r(n, nfeat, winlen), "b": r(n) if bias else None, "stride": stride, "padding": pad,
         "activation": conv_act},
        {"type": "reverse", "sublayer": gru(n)},
        gru(n),
        {"type": "feed-forward", "W": r(n, n), "b": r(n) if bias else None, "activation": "tanh"},
        {"type": "parallel", "sublayers": [gru(n), {"type": "reverse", "sublayer": gru(n)}]},           # a birnn
        {"type": "feed-forward", "W": r(n, 2 * n), "b": r(n) if bias else None, "activation": "tanh"},
        {"type": "reverse", "sublayer": gru(n)},
        {"type": "softmax", "W": r(nstate, n), "b": r(nstate) if bias else None}]}


@pytest.mark.parametrize("min_prob,l2,drop,conv_act,bias", [(0.0, 0.0, 0, "elu", True), (1e-3, 0.01, 2, "tanh", True),
                                                            (1e-30, 0.0, 1, "relu", False)])
def test_gradients_match_finite_differences(min_prob, l2, drop, conv_act, bias):
    rs = np.random.RandomState(5)
    spec = _net(rs, conv_act=conv_act, bias=bias)
    T, B = 22, 3
    x = rs.normal(size=(T, B, 1))
    Tout = (T + 4 - 5) // 2 + 1
    labels = rs.randint(0, 9, size=(Tout, B))
    weights = rs.uniform(0.5, 1.5, size=(Tout, B))
    loss, acc, grads = ot.loss_and_grads(spec, x, labels, weights, min_prob, l2, drop)
    loss2, acc2 = ot.loss_only(spec, x, labels, weights, min_prob, l2, drop)
    assert loss == pytest.approx(loss2, rel=1e-12) and acc == acc2
    params = ot.params_of(spec)
    assert len(params) == len(grads) and all(np.shape(p) == g.shape for p, g in zip(params, grads))
    eps = 1e-6
    for p, g in zip(params, grads):
        flat = p.reshape(-1)                       # a view: perturbing it perturbs the network
        for idx in rs.choice(flat.size, size=min(6, flat.size), replace=False):
            keep = flat[idx]
            flat[idx] = keep + eps
            up, _ = ot.loss_only(spec, x, labels, weights, min_prob, l2, drop)
            flat[idx] = keep - eps
            dn, _ = ot.loss_only(spec, x, labels, weights, min_prob, l2, drop)
            flat[idx] = keep
            assert g.reshape(-1)[idx] == pytest.approx((up - dn) / (2 * eps), rel=2e-5, abs=1e-8)


def test_window_network_gradients():
    """Event-feature front end (models/baseline_gru.py): Window, then layers; Window in the middle exercises its reverse pass."""
    rs = np.random.RandomState(9)
    r = lambda *shape: rs.normal(size=shape) * 0.5
    n = 5
    gru = lambda i: {"type": "GRU", "iW": r(3 * n, i), "sW": r(2 * n, n), "sW2": r(n, n), "b": r(3 * n), "activation": "tanh",
                     "gate": "sigmoid"}
    spec = {"type": "serial", "sublayers": [{"type": "window", "w": 3}, gru(12), {"type": "window", "w": 5},
                                            {"type": "feed-forward", "W": r(n, 5 * n), "b": r(n), "activation": "tanh"},
                                            {"type": "softmax", "W": r(7, n), "b": r(7)}]}
    x = rs.normal(size=(15, 2, 4))
    labels = rs.randint(0, 7, size=(15, 2))
    weights = rs.uniform(0.5, 1.5, size=(15, 2))
    loss, acc, grads = ot.loss_and_grads(spec, x, labels, weights, 1e-4, 0.0, 1)
    eps = 1e-6
    for p, g in zip(ot.params_of(spec), grads):
        flat = p.reshape(-1)
        for idx in rs.choice(flat.size, size=min(5, flat.size), replace=False):
            keep = flat[idx]
            flat[idx] = keep + eps
            up, _ = ot.loss_only(spec, x, labels, weights, 1e-4, 0.0, 1)
            flat[idx] = keep - eps
            dn, _ = ot.loss_only(spec, x, labels, weights, 1e-4, 0.0, 1)
            flat[idx] = keep
            assert g.reshape(-1)[idx] == pytest.approx((up - dn) / (2 * eps), rel=2e-5, abs=1e-8)


@pytest.mark.parametrize("bias,peep", [(True, True), (False, False)])
def test_lstm_network_gradients(bias, peep):
    """models/baseline_lstm.py's structure in small: Window, birnn of peephole Lstm cells, FeedForward, Softmax."""
    rs = np.random.RandomState(21)
    r = lambda *shape: rs.normal(size=shape) * 0.5
    n = 4
    lstm = lambda i: {"type": "LSTM", "iW": r(4 * n, i), "sW": r(4 * n, n), "b": r(4 * n) if bias else None,
                      "p": r(3, n) if peep else None, "activation": "tanh", "gate": "sigmoid"}
    spec = {"type": "serial", "sublayers": [
        {"type": "window", "w": 3},
        {"type": "parallel", "sublayers": [lstm(6), {"type": "reverse", "sublayer": lstm(6)}]},
        {"type": "feed-forward", "W": r(n, 2 * n), "b": r(n), "activation": "tanh"},
        {"type": "reverse", "sublayer": lstm(n)},
        {"type": "softmax", "W": r(5, n), "b": r(5)}]}
    x = rs.normal(size=(14, 3, 2))
    labels = rs.randint(0, 5, size=(14, 3))
    weights = rs.uniform(0.5, 1.5, size=(14, 3))
    loss, acc, grads = ot.loss_and_grads(spec, x, labels, weights, 1e-4, 0.0, 1)
    assert loss == pytest.approx(ot.loss_only(spec, x, labels, weights, 1e-4, 0.0, 1)[0], rel=1e-12)
    params = ot.params_of(spec)
    assert len(params) == len(grads)
    eps = 1e-6
    for p, g in zip(params, grads):
        assert np.shape(p) == g.shape
        flat = p.reshape(-1)
        for idx in rs.choice(flat.size, size=min(6, flat.size), replace=False):
            keep = flat[idx]
            flat[idx] = keep + eps
            up, _ = ot.loss_only(spec, x, labels, weights, 1e-4, 0.0, 1)
            flat[idx] = keep - eps
            dn, _ = ot.loss_only(spec, x, labels, weights, 1e-4, 0.0, 1)
            flat[idx] = keep
            assert g.reshape(-1)[idx] == pytest.approx((up - dn) / (2 * eps), rel=2e-5, abs=1e-8)


def test_adamski_first_steps_closed_form():
    """updates.py:36-89: with momentum/variance starting at zero, step 1 moves every parameter by
    lr_1 * (1-d1) g / (sqrt((1-d2) g^2) + eps) with lr_1 = rate sqrt(1-d2) / momentum_factor_1."""
    p = [np.array([1.0, -2.0, 3.0], np.float32)]
    g = [np.array([0.5, -7.0, 1e-3], np.float32)]            # -7 is clipped to -5
    opt = ot.Adamski(p, decay=(0.9, 0.999))
    new = opt.step(p, g, rate=1e-3)
    d1, d2, mr = 0.9, 0.999, 0.0005
    mk = (1 - d1) * d1 * np.exp(-mr) / (1 - np.exp(-mr) * d1)
    mf = mk * np.expm1(0.0) - np.expm1(np.log(d1))          # t = 0 in the first term
    lr = 1e-3 * np.sqrt(-np.expm1(np.log(d2))) / mf
    gc = np.clip(g[0].astype(np.float64), -5, 5)
    want = p[0] - lr * ((1 - d1) * gc) / (np.sqrt((1 - d2) * gc ** 2) + 1e-8)
    np.testing.assert_allclose(new[0], want, rtol=2e-5)
    assert opt.t == 1.0
    # momentum is phased in: the decay applied to the old momentum at step k is d1 (1 - exp(-k mrate))
    _, md = opt.scalars(1e-3)
    assert md == pytest.approx(d1 * (1 - np.exp(-2 * mr)), rel=1e-4)
    # mrate=None gives plain ADAM's bias-corrected step size
    adam = ot.Adamski(p, decay=(0.9, 0.999), mrate=None)
    lr1, md1 = adam.scalars(1e-3)
    assert md1 == pytest.approx(0.9, rel=1e-6)
    assert lr1 == pytest.approx(1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9), rel=1e-4)
