"""Second, independent restatement of the reference's layer maths in numpy (float64 by default).

TEST INFRASTRUCTURE ONLY (see oracle/sloika_oracle.c header).  Written as a literal transcription of
the Theano expressions in sloika/layers.py so that the C oracle and the HIP kernels can both be
checked against "the formula as the reference wrote it", evaluated in float64.

Pinned (tests/test_oracle_reference_layers.py, <= 1e-10) to what the reference's own sloika/layers.py, conv.py,
activation.py and models/*.py compute when executed unmodified under the eager Theano stand-in of
tests/golden/theano_standin (fixtures tests/golden/layers.npz): the formulas here ARE the reference's, not merely close.
Theano's own float32 kernels were never run (not installable); FeedForward / Softmax / Window additionally restate
the numpy known-answer tests of test/unit/test_layers.py:58-69, 118-125, 246-266.
"""
import numpy as np
from scipy.special import erf as _erf


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _relu(x):
    return np.where(x > 0, x, 0.0)


ACT = {                                                       # sloika/activation.py
    "linear": lambda x: x,                                    # :8-9
    "relu": _relu,                                            # :12-13
    "relu_smooth": lambda x: (lambda y: np.square(y) - 2.0 * y + x + np.abs(x))(np.clip(x, 0.0, 1.0)),  # :16-18
    "softplus": lambda x: _relu(x) + np.log1p(np.exp(-np.abs(x))),   # :21-35
    "elu": lambda x: np.where(x > 0, x, np.expm1(np.minimum(x, 0.0))),  # :38-42
    "exp": np.exp,                                            # :45-46
    "tanh": np.tanh,                                          # :52-53
    "sigmoid": _sigmoid,                                      # :56-57
    "erf": _erf,                                              # :60-61
    "L1mL2": lambda x: x / np.sqrt(1.0 + 0.5 * np.square(x)),  # :64-65
    "fair": lambda x: x / (1.0 + np.abs(x) / 1.3998),         # :68-69
    "retu": lambda x: np.tanh(_relu(x)),                      # :72-78
    "tanh_pm": lambda x: np.clip(x, -1.0, 1.0),               # :81-85
    "sigmoid_pm": lambda x: np.clip(0.5 + 0.25 * x, 0.0, 1.0),  # :88-92
    "bounded_linear": lambda x: np.clip(x, -1.0, 1.0),        # :95-98
    "sin": np.sin,                                            # :102-103
    "cauchy": lambda x: x / (1.0 + np.square(x / 2.3849)),    # :106-107
    "geman_mcclure": lambda x: x / np.square(1.0 + np.square(x)),  # :110-111
    "welsh": lambda x: x * np.exp(-np.square(x / 2.9846)),    # :114-115
}


def conv1d(x, W, b, stride, padding, act, dtype=np.float64):
    """layers.py:417-419, conv.py:66-77,90-111"""
    x, W = np.asarray(x, dtype), np.asarray(W, dtype)
    T, B, Cin = x.shape
    Cout, _, winlen = W.shape
    xp = np.concatenate([np.zeros((padding[0], B, Cin), dtype), x, np.zeros((padding[1], B, Cin), dtype)], 0)
    Tout = (xp.shape[0] - winlen) // stride + 1
    y = np.zeros((Tout, B, Cout), dtype)
    for k in range(winlen):
        seg = xp[k: k + (Tout - 1) * stride + 1: stride]            # [Tout, B, Cin]
        y += np.tensordot(seg, W[:, :, k], axes=(2, 1))
    if b is not None:
        y = y + np.asarray(b, dtype)
    return ACT[act](y)


def window(x, w):
    """layers.py:346-351 (literal)"""
    T, B, F = x.shape
    zeros = np.zeros((w // 2, B, F), x.dtype)
    pad = np.concatenate([zeros, x, zeros], axis=0)
    tmp = np.concatenate([pad[i: 1 + i - w] for i in range(w - 1)], axis=2)
    return np.concatenate([tmp, pad[w - 1:]], axis=2)


def feedforward(x, W, b, act, dtype=np.float64):
    """layers.py:157-158"""
    y = np.tensordot(np.asarray(x, dtype), np.asarray(W, dtype), axes=(2, 1))
    if b is not None:
        y = y + np.asarray(b, dtype)
    return ACT[act](y)


def softmax(x, W, b, dtype=np.float64):
    """layers.py:309-314"""
    tmp = np.tensordot(np.asarray(x, dtype), np.asarray(W, dtype), axes=(2, 1))
    if b is not None:
        tmp = tmp + np.asarray(b, dtype)
    m = np.max(tmp, axis=2, keepdims=True)
    out = np.exp(tmp - m)
    return out / np.sum(out, axis=2, keepdims=True)


def gru(x, iW, sW, sW2, b, act="tanh", gate="sigmoid", reverse=False, dtype=np.float64):
    """layers.py:1010-1021 (step), :85-88 (scan, zero state), :1449-1450 (Reverse)"""
    x, iW, sW, sW2 = (np.asarray(a, dtype) for a in (x, iW, sW, sW2))
    T, B, _ = x.shape
    n = sW2.shape[0]
    bb = np.zeros(3 * n, dtype) if b is None else np.asarray(b, dtype)
    xs = x[::-1] if reverse else x
    state = np.zeros((B, n), dtype)
    out = np.empty((T, B, n), dtype)
    for t in range(T):
        vI = np.tensordot(xs[t], iW, axes=(1, 1)) + bb
        vS = np.tensordot(state, sW, axes=(1, 1))
        vT = (vI[:, :2 * n] + vS).reshape((-1, 2, n))
        z = ACT[gate](vT[:, 0])
        r = ACT[gate](vT[:, 1])
        y = np.tensordot(r * state, sW2, axes=(1, 1))
        hbar = ACT[act](vI[:, 2 * n:] + y)
        state = z * state + (1 - z) * hbar
        out[t] = state
    return out[::-1] if reverse else out


def lstm(x, iW, sW, b, p, act="tanh", gate="sigmoid", reverse=False, dtype=np.float64):
    """layers.py:677-697 (interleaved gate layout via reshape((-1, size, 4)))"""
    x, iW, sW = (np.asarray(a, dtype) for a in (x, iW, sW))
    T, B, _ = x.shape
    n = sW.shape[1]
    bb = np.zeros(4 * n, dtype) if b is None else np.asarray(b, dtype)
    pp = np.zeros((3, n), dtype) if p is None else np.asarray(p, dtype)
    xs = x[::-1] if reverse else x
    in_state = np.zeros((B, 2 * n), dtype)
    out_all = np.empty((T, B, n), dtype)
    for t in range(T):
        vW = np.tensordot(xs[t], iW, axes=(1, 1))
        out_prev = in_state[:, :n]
        state = in_state[:, n:]
        outW = np.tensordot(out_prev, sW, axes=(1, 1))
        sumW = (vW + outW + bb).reshape((-1, n, 4))
        out_state = state * ACT[gate](sumW[:, :, 2] + state * pp[1])
        out_state = out_state + ACT[act](sumW[:, :, 0]) * ACT[gate](sumW[:, :, 1] + state * pp[0])
        out = ACT[act](out_state) * ACT[gate](sumW[:, :, 3] + out_state * pp[2])
        in_state = np.concatenate((out, out_state), axis=1)
        out_all[t] = out
    return out_all[::-1] if reverse else out_all


def run_network(spec, x, dtype=np.float64):
    t = spec["type"]
    if t == "serial":
        for sub in spec["sublayers"]:
            x = run_network(sub, x, dtype)
        return x
    if t == "parallel":
        return np.concatenate([run_network(sub, x, dtype) for sub in spec["sublayers"]], axis=2)
    if t == "reverse":
        return run_network(spec["sublayer"], x[::-1], dtype)[::-1]
    if t == "GRU":
        return gru(x, spec["iW"], spec["sW"], spec["sW2"], spec.get("b"), spec["activation"], spec["gate"],
                   dtype=dtype)
    if t == "LSTM":
        return lstm(x, spec["iW"], spec["sW"], spec.get("b"), spec.get("p"), spec["activation"], spec["gate"],
                    dtype=dtype)
    if t == "convolution":
        return conv1d(x, spec["W"], spec.get("b"), spec["stride"], tuple(spec["padding"]), spec["activation"], dtype)
    if t == "window":
        return window(np.asarray(x, dtype), spec["w"])
    if t == "feed-forward":
        return feedforward(x, spec["W"], spec.get("b"), spec["activation"], dtype)
    if t == "softmax":
        return softmax(x, spec["W"], spec.get("b"), dtype)
    raise ValueError("oracle_np: unsupported layer type %r" % t)


def viterbi_py(post, klen, skip_pen=0.0, log=False, nbase=4):
    """Scalar python loop form of decode.py:39-93, for tiny cases (third opinion on tie-breaks)."""
    post = np.asarray(post)
    nev, nst = post.shape
    nkmer = nbase ** klen
    lpost = np.log(post + 1e-10) if not log else post
    typ = lpost.dtype.type
    v = [typ(a) for a in lpost[0][1:]]
    tb = [[0] * nkmer for _ in range(nev)]
    r1, r2 = nkmer // nbase, nkmer // (nbase * nbase)
    for i in range(1, nev):
        pv, v = v, [typ(0)] * nkmer
        for s in range(nkmer):
            j1, j2 = s // nbase, s // (nbase * nbase)
            c1 = [pv[a * r1 + j1] for a in range(nbase)]
            c2 = [pv[a * r2 + j2] for a in range(nbase * nbase)]
            a1 = max(range(nbase), key=lambda a: (c1[a], -a))
            a2 = max(range(nbase * nbase), key=lambda a: (c2[a], -a))
            sstep, sskip = c1[a1], typ(c2[a2] - typ(skip_pen))
            nv = typ(lpost[i][1 + s] + max(sstep, sskip))
            frm = a1 * r1 + j1 if sstep > sskip else a2 * r2 + j2
            stay = typ(pv[s] + lpost[i][0])
            tb[i][s] = frm if nv > stay else -1
            v[s] = nv if nv > stay else stay
    best = max(range(nkmer), key=lambda s: (v[s], -s))
    seq = [best]
    for i in range(nev - 1, 0, -1):
        ts = tb[i][seq[-1]]
        if ts >= 0:
            seq.append(ts)
    return v[best], seq[::-1]
