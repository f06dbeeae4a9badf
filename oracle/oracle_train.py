"""CPU restatement (numpy, float64) of the reference's TRAINING step -- SURVEY.md section 8 row f2.

TEST INFRASTRUCTURE ONLY (see oracle/sloika_oracle.c header): imported by tests/ and nothing else.

What is restated, and from where:
  * the loss and accuracy of bin/train_network.py:124-142 (`wrap_network`):
        post  = min_prob + (1 - min_prob) * network.run(x)
        loss  = l2 * param_sqr(network) + mean((weights * -log post[t, b, labels[t, b]])[drop : -drop])
        acc   = mean((argmax(post, axis=2) == labels)[drop : -drop])
  * the gradient th.grad(loss, network.params()) (sloika/updates.py:66) -- the reference gets it from Theano's
    automatic differentiation; here it is the hand-derived reverse pass of the layer formulas restated in
    oracle/oracle_np.py (Convolution layers.py:417-419, Gru.step layers.py:1010-1021, Lstm.step layers.py:677-697,
    Softmax layers.py:309-314,
    FeedForward layers.py:157-158, Window layers.py:346-351, Reverse layers.py:1449-1450, Parallel layers.py:1486-1487, Serial layers.py:1500-1504);
  * the "ADAMski" update sloika/updates.py:36-89 (float32 arithmetic like the reference's shared variables), `sgd`
    updates.py:9-33 and `param_sqr` updates.py:92-103.

Pinned (tests/test_oracle_reference_layers.py::test_training_step_vs_reference_code) to the reference's own
bin/train_network.py:wrap_network + sloika/updates.py:adam, imported unmodified and executed under the eager Theano
stand-in of tests/golden/theano_standin, whose `theano.grad` is automatic differentiation of the reference's own graph:
loss and accuracy to 1e-9, every gradient to 1e-9 of its largest entry, the parameters after two or three ADAMski steps to
2e-6 (float32 optimiser arithmetic), on three small networks covering Convolution, Gru, birnn, FeedForward, Window, Lstm.
Not pinned: Theano's own kernels (never run).  Independently of that: (1) the forward pass is oracle_np.run_network;
(2) tests/test_oracle_train.py checks every gradient against central finite differences in float64; (3) the update rule is
checked against the closed form of its first steps.
"""
import numpy as np

from . import oracle_np as onp


def _dact(name, y, a):
    """Derivative of activation `name` given its output y and its argument a."""
    if name == "tanh":
        return 1.0 - y * y
    if name == "sigmoid":
        return y * (1.0 - y)
    if name == "linear":
        return np.ones_like(y)
    if name == "relu":
        return (a > 0).astype(y.dtype)
    if name == "elu":
        return np.where(a > 0, 1.0, y + 1.0)
    raise ValueError("oracle_train: no derivative restated for activation %r" % name)


def params_of(spec):
    """Parameter arrays in network.params() order (layers.py: Serial/Reverse concatenate their sublayers')."""
    t = spec["type"]
    if t == "serial":
        return [p for sub in spec["sublayers"] for p in params_of(sub)]
    if t == "reverse":
        return params_of(spec["sublayer"])
    if t == "parallel":
        return [p for sub in spec["sublayers"] for p in params_of(sub)]
    if t == "GRU":
        return [spec[k] for k in ("iW", "sW", "sW2", "b") if spec.get(k) is not None]
    if t == "LSTM":                                                # layers.py Lstm.params(): iW, sW, b, p
        return [spec[k] for k in ("iW", "sW", "b", "p") if spec.get(k) is not None]
    if t in ("convolution", "softmax", "feed-forward"):
        return [spec[k] for k in ("W", "b") if spec.get(k) is not None]
    if t == "window":
        return []
    raise ValueError("oracle_train: unsupported layer type %r" % t)


def _forward(spec, x):
    """Returns (output, tape) with everything the reverse pass needs."""
    t = spec["type"]
    f64 = np.float64
    if t == "serial":
        tapes = []
        for sub in spec["sublayers"]:
            x, tp = _forward(sub, x)
            tapes.append(tp)
        return x, tapes
    if t == "reverse":
        y, tp = _forward(spec["sublayer"], x[::-1])
        return y[::-1], tp
    if t == "parallel":                                            # layers.py:1486-1487
        outs = [_forward(sub, x) for sub in spec["sublayers"]]
        return np.concatenate([o for o, _ in outs], axis=2), ([tp for _, tp in outs], [o.shape[2] for o, _ in outs])
    if t == "window":                                              # layers.py:346-351
        return onp.window(x, spec["w"]), (x.shape, spec["w"])
    if t == "convolution":
        W = np.asarray(spec["W"], f64)
        pad, stride = tuple(spec["padding"]), spec["stride"]
        T, B, Cin = x.shape
        winlen = W.shape[2]
        xp = np.concatenate([np.zeros((pad[0], B, Cin)), x, np.zeros((pad[1], B, Cin))], 0)
        Tout = (xp.shape[0] - winlen) // stride + 1
        a = np.zeros((Tout, B, W.shape[0]))
        for k in range(winlen):
            a += np.tensordot(xp[k: k + (Tout - 1) * stride + 1: stride], W[:, :, k], axes=(2, 1))
        if spec.get("b") is not None:
            a = a + np.asarray(spec["b"], f64)
        y = onp.ACT[spec["activation"]](a)
        return y, (xp, a, y, Tout)
    if t == "feed-forward":
        a = np.tensordot(x, np.asarray(spec["W"], f64), axes=(2, 1))
        if spec.get("b") is not None:
            a = a + np.asarray(spec["b"], f64)
        y = onp.ACT[spec["activation"]](a)
        return y, (x, a, y)
    if t == "softmax":
        y = onp.softmax(x, spec["W"], spec.get("b"))
        return y, (x, y)
    if t == "GRU":
        iW, sW, sW2 = (np.asarray(spec[k], f64) for k in ("iW", "sW", "sW2"))
        n = sW2.shape[0]
        b = np.zeros(3 * n) if spec.get("b") is None else np.asarray(spec["b"], f64)
        T, B, _ = x.shape
        h = np.zeros((B, n))
        out = np.empty((T, B, n))
        tape = []
        for s in range(T):
            vI = x[s] @ iW.T + b
            vS = h @ sW.T
            az, ar = vI[:, :n] + vS[:, :n], vI[:, n:2 * n] + vS[:, n:]
            z, r = onp.ACT[spec["gate"]](az), onp.ACT[spec["gate"]](ar)
            ac = vI[:, 2 * n:] + (r * h) @ sW2.T
            c = onp.ACT[spec["activation"]](ac)
            tape.append((h, z, r, c, az, ar, ac))
            h = z * h + (1 - z) * c
            out[s] = h
        return out, (x, tape)
    if t == "LSTM":                                                # layers.py:677-697 (gate rows interleaved: j*4 + gate)
        iW, sW = np.asarray(spec["iW"], f64), np.asarray(spec["sW"], f64)
        n = sW.shape[1]
        b = np.zeros(4 * n) if spec.get("b") is None else np.asarray(spec["b"], f64)
        pp = np.zeros((3, n)) if spec.get("p") is None else np.asarray(spec["p"], f64)
        T, B, _ = x.shape
        out, cell = np.zeros((B, n)), np.zeros((B, n))
        outs = np.empty((T, B, n))
        tape = []
        for s in range(T):
            sm = (x[s] @ iW.T + out @ sW.T + b).reshape((-1, n, 4))
            g = onp.ACT[spec["activation"]](sm[:, :, 0])
            a_i, a_f = sm[:, :, 1] + cell * pp[0], sm[:, :, 2] + cell * pp[1]
            i, f = onp.ACT[spec["gate"]](a_i), onp.ACT[spec["gate"]](a_f)
            cn = cell * f + g * i
            a_o = sm[:, :, 3] + cn * pp[2]
            o = onp.ACT[spec["gate"]](a_o)
            tc = onp.ACT[spec["activation"]](cn)
            tape.append((out, cell, cn, g, i, f, o, tc, sm[:, :, 0], a_i, a_f, a_o))
            out, cell = tc * o, cn
            outs[s] = out
        return outs, (x, tape)
    raise ValueError("oracle_train: unsupported layer type %r" % t)


def _backward(spec, tape, dy):
    """Returns (dx, [parameter gradients in params_of order])."""
    t = spec["type"]
    f64 = np.float64
    if t == "serial":
        grads = []
        for sub, tp in zip(reversed(spec["sublayers"]), reversed(tape)):
            dy, g = _backward(sub, tp, dy)
            grads = g + grads
        return dy, grads
    if t == "reverse":
        dx, g = _backward(spec["sublayer"], tape, dy[::-1])
        return dx[::-1], g
    if t == "parallel":
        tapes, sizes = tape
        dx, grads, off = 0.0, [], 0
        for sub, tp, size in zip(spec["sublayers"], tapes, sizes):
            d, g = _backward(sub, tp, dy[:, :, off:off + size])
            dx, grads, off = dx + d, grads + g, off + size
        return dx, grads
    if t == "window":
        (T, B, F), w = tape                                        # out[t, :, j*F:(j+1)*F] = xpad[t + j], xpad = w//2 zeros each side
        dxp = np.zeros((T + 2 * (w // 2), B, F))
        for j in range(w):
            dxp[j:j + T] += dy[:, :, j * F:(j + 1) * F]
        return dxp[w // 2: w // 2 + T], []
    if t == "convolution":
        xp, a, y, Tout = tape
        W = np.asarray(spec["W"], f64)
        pad, stride = tuple(spec["padding"]), spec["stride"]
        da = dy * _dact(spec["activation"], y, a)
        dW = np.zeros_like(W)
        dxp = np.zeros_like(xp)
        for k in range(W.shape[2]):
            seg = slice(k, k + (Tout - 1) * stride + 1, stride)
            dW[:, :, k] = np.tensordot(da, xp[seg], axes=([0, 1], [0, 1]))
            dxp[seg] += np.tensordot(da, W[:, :, k], axes=(2, 0))
        g = [dW] + ([da.sum(axis=(0, 1))] if spec.get("b") is not None else [])
        return dxp[pad[0]: xp.shape[0] - pad[1]], g
    if t == "feed-forward":
        x, a, y = tape
        da = dy * _dact(spec["activation"], y, a)
        g = [np.tensordot(da, x, axes=([0, 1], [0, 1]))] + ([da.sum(axis=(0, 1))] if spec.get("b") is not None else [])
        return np.tensordot(da, np.asarray(spec["W"], f64), axes=(2, 0)), g
    if t == "softmax":
        x, y = tape
        da = y * (dy - np.sum(dy * y, axis=2, keepdims=True))
        g = [np.tensordot(da, x, axes=([0, 1], [0, 1]))] + ([da.sum(axis=(0, 1))] if spec.get("b") is not None else [])
        return np.tensordot(da, np.asarray(spec["W"], f64), axes=(2, 0)), g
    if t == "GRU":
        x, steps = tape
        iW, sW, sW2 = (np.asarray(spec[k], f64) for k in ("iW", "sW", "sW2"))
        n = sW2.shape[0]
        T, B, _ = x.shape
        diW, dsW, dsW2, db = np.zeros_like(iW), np.zeros_like(sW), np.zeros_like(sW2), np.zeros(3 * n)
        dx = np.empty_like(x)
        dh = np.zeros((B, n))
        for s in range(T - 1, -1, -1):
            h, z, r, c, az, ar, ac = steps[s]
            g = dy[s] + dh
            dac = g * (1 - z) * _dact(spec["activation"], c, ac)
            daz = g * (h - c) * _dact(spec["gate"], z, az)
            drh = dac @ sW2                                       # gradient of (r * h)
            dar = drh * h * _dact(spec["gate"], r, ar)
            da = np.concatenate([daz, dar, dac], axis=1)          # gradient of vI (and of vS for the first 2n)
            dh = g * z + drh * r + da[:, :2 * n] @ sW
            diW += da.T @ x[s]
            dsW += da[:, :2 * n].T @ h
            dsW2 += dac.T @ (r * h)
            db += da.sum(axis=0)
            dx[s] = da @ iW
        g = [diW, dsW, dsW2] + ([db] if spec.get("b") is not None else [])
        return dx, g
    if t == "LSTM":
        x, steps = tape
        iW, sW = np.asarray(spec["iW"], f64), np.asarray(spec["sW"], f64)
        n = sW.shape[1]
        pp = np.zeros((3, n)) if spec.get("p") is None else np.asarray(spec["p"], f64)
        T, B, _ = x.shape
        diW, dsW, db, dp = np.zeros_like(iW), np.zeros_like(sW), np.zeros(4 * n), np.zeros((3, n))
        dx = np.empty_like(x)
        d_out, d_cell = np.zeros((B, n)), np.zeros((B, n))
        for s in range(T - 1, -1, -1):
            out_prev, c_prev, cn, g, i, f, o, tc, a_g, a_i, a_f, a_o = steps[s]
            go = dy[s] + d_out
            do_pre = go * tc * _dact(spec["gate"], o, a_o)
            dc = go * o * _dact(spec["activation"], tc, cn) + do_pre * pp[2] + d_cell
            di_pre = dc * g * _dact(spec["gate"], i, a_i)
            df_pre = dc * c_prev * _dact(spec["gate"], f, a_f)
            dg_pre = dc * i * _dact(spec["activation"], g, a_g)
            d_cell = dc * f + di_pre * pp[0] + df_pre * pp[1]
            dsum = np.stack([dg_pre, di_pre, df_pre, do_pre], axis=2).reshape(B, 4 * n)      # rows j*4 + gate
            d_out = dsum @ sW
            diW += dsum.T @ x[s]
            dsW += dsum.T @ out_prev
            db += dsum.sum(axis=0)
            dp += np.stack([(di_pre * c_prev).sum(0), (df_pre * c_prev).sum(0), (do_pre * cn).sum(0)])
            dx[s] = dsum @ iW
        g_list = [diW, dsW] + ([db] if spec.get("b") is not None else []) + ([dp] if spec.get("p") is not None else [])
        return dx, g_list
    raise ValueError("oracle_train: unsupported layer type %r" % t)


def loss_only(spec, x, labels, weights, min_prob=0.0, l2=0.0, drop=0):
    """train_network.py:124-137, forward only (oracle_np.run_network): returns (loss, acc)."""
    post = min_prob + (1.0 - min_prob) * onp.run_network(spec, np.asarray(x, np.float64))
    T, B, _ = post.shape
    sl = slice(drop, None if drop == 0 else -drop)
    tt, bb = np.meshgrid(np.arange(T), np.arange(B), indexing="ij")
    lpe = -np.log(post[tt, bb, labels])
    penalty = l2 * sum(float(np.sum(np.square(np.asarray(p, np.float64)))) for p in params_of(spec))
    loss = penalty + np.mean((np.asarray(weights, np.float64) * lpe)[sl])
    acc = np.mean((np.argmax(post, axis=2) == labels)[sl])
    return float(loss), float(acc)


def loss_and_grads(spec, x, labels, weights, min_prob=0.0, l2=0.0, drop=0):
    """Loss, accuracy (train_network.py:124-137) and d loss / d params in network.params() order (updates.py:66)."""
    x = np.asarray(x, np.float64)
    y, tape = _forward(spec, x)
    post = min_prob + (1.0 - min_prob) * y
    T, B, _ = post.shape
    lo, hi = drop, (T if drop == 0 else T - drop)
    tt, bb = np.meshgrid(np.arange(T), np.arange(B), indexing="ij")
    p_lab = post[tt, bb, labels]
    w = np.asarray(weights, np.float64)
    mask = np.zeros((T, B))
    mask[lo:hi] = 1.0 / ((hi - lo) * B)
    params = params_of(spec)
    loss = l2 * sum(float(np.sum(np.square(np.asarray(p, np.float64)))) for p in params) + np.sum(mask * w * -np.log(p_lab))
    acc = np.mean((np.argmax(post, axis=2) == labels)[lo:hi])
    dy = np.zeros_like(y)
    dy[tt, bb, labels] = -(mask * w) * (1.0 - min_prob) / p_lab
    _, grads = _backward(spec, tape, dy)
    grads = [g + 2.0 * l2 * np.asarray(p, np.float64) for g, p in zip(grads, params)]
    return float(loss), float(acc), grads


class Adamski(object):
    """sloika/updates.py:36-89, evaluated in float32 like the reference's shared variables (sloika_dtype)."""

    def __init__(self, params, decay=(0.9, 0.999), epsilon=1e-8, clip=5.0, mrate=0.0005):
        f32 = np.float32
        self.decay, self.eps, self.clip = (f32(decay[0]), f32(decay[1])), f32(epsilon), f32(clip)
        if mrate is not None:                                                    # :54-58
            self.m_rate = -f32(mrate)
            m_p = np.exp(self.m_rate)
            self.m_k = f32((1.0 - decay[0]) * decay[0] * m_p / (1.0 - m_p * decay[0]))
        else:                                                                    # :59-62
            self.m_rate, self.m_k = -f32(1e30), f32(0.0)
        self.ldecay = np.log(np.array(decay), dtype=np.float32)                  # :68
        self.t = f32(0.0)
        self.momentum = [np.zeros(np.shape(p), f32) for p in params]
        self.variance = [np.zeros(np.shape(p), f32) for p in params]

    def scalars(self, rate):
        """(lr_t, momentum_decay) of this step; advances t  (:73-76)."""
        f32 = np.float32
        t_new = f32(self.t + f32(1.0))
        with np.errstate(over="ignore"):
            momentum_factor = f32(self.m_k * np.expm1(f32(self.t * f32(self.ldecay[0] + self.m_rate)))
                                  - np.expm1(f32(t_new * self.ldecay[0])))
            lr_t = f32(f32(rate) * np.sqrt(-np.expm1(f32(t_new * self.ldecay[1]))) / momentum_factor)
            momentum_decay = f32(-self.decay[0] * np.expm1(f32(t_new * self.m_rate)))
        self.t = t_new
        return lr_t, momentum_decay

    def step(self, params, grads, rate):
        """Returns the updated parameters (float32)  (:77-87)."""
        f32 = np.float32
        lr_t, momentum_decay = self.scalars(rate)
        out = []
        for i, (p, g) in enumerate(zip(params, grads)):
            gc = np.clip(np.asarray(g, f32), -self.clip, self.clip)
            self.momentum[i] = momentum_decay * self.momentum[i] + (f32(1.0) - self.decay[0]) * gc
            self.variance[i] = self.decay[1] * self.variance[i] + (f32(1.0) - self.decay[1]) * np.square(gc)
            out.append((np.asarray(p, f32) - lr_t * self.momentum[i] / (np.sqrt(self.variance[i]) + self.eps)).astype(f32))
        return out
