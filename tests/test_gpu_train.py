"""Parity of the training step (sloika_amd/train.py + csrc/train.hip through the C ABI)
with the float64 oracle of bin/train_network.py:124-142 and sloika/updates.py:36-89 (oracle/oracle_train.py, itself
checked against finite differences in tests/test_oracle_train.py)."""
import os

import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu


def _build(rs, n=32, nstate=17, winlen=5, stride=2, nlayer=3, conv=True, bias=True, scale=0.5):
    from sloika_amd import activation, layers
    init = lambda shape: (rs.normal(size=shape) * scale).astype(np.float32)
    subs = []
    if conv:
        subs.append(layers.Convolution(1, n, winlen, stride, init=init, has_bias=bias, fun=activation.elu))
    for l in range(nlayer):
        g = layers.Gru(n, n, init=init, has_bias=bias, fun=activation.tanh)
        subs.append(layers.Reverse(g) if l % 2 == 0 else g)
    subs.append(layers.Softmax(n, nstate, init=init, has_bias=bias))
    return layers.Serial(subs)


def _batch(rs, net, T, B, nfeat=1):
    x = rs.normal(size=(T, B, nfeat)).astype(np.float32)
    first = net.layers[0]
    To = first.out_len(T) if hasattr(first, "out_len") else T
    labels = rs.randint(0, net.size, size=(To, B)).astype(np.int32)
    weights = rs.uniform(0.5, 1.5, size=(To, B)).astype(np.float32)
    return x, labels, weights


def _assert_grads_close(got, want, tol=2e-4):
    """Relative to the largest entry of each tensor: float32 sums over T*B rows against float64."""
    assert len(got) == len(want)
    for k, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape, k
        scale = max(float(np.abs(w).max()), 1e-6)
        np.testing.assert_allclose(g / scale, w / scale, atol=tol, err_msg="parameter %d" % k)


@pytest.mark.parametrize("n,nstate,T,B,min_prob,l2,drop,bias", [
    (32, 17, 61, 5, 0.0, 0.0, 0, True),
    (32, 17, 40, 3, 1e-3, 0.01, 3, True),
    (16, 5, 24, 2, 1e-30, 0.0, 1, False),
    (96, 1025, 75, 4, 1e-30, 0.0, 2, True),        # the shapes of models/raw_0.98_rgrgr.py
    (64, 260, 2100, 2, 1e-5, 0.0, 20, True),       # more rows than one slice of the A^T B contraction
])
def test_loss_and_gradients_vs_oracle(n, nstate, T, B, min_prob, l2, drop, bias):
    need_gpu()
    from oracle import oracle_train as ot
    from sloika_amd import train
    rs = np.random.RandomState(n + T)
    net = _build(rs, n=n, nstate=nstate, stride=5 if n == 96 else 2, winlen=11 if n == 96 else 5, nlayer=5 if n == 96 else 3,
                 bias=bias, scale=0.5 if T < 1000 else 0.3)
    x, labels, weights = _batch(rs, net, T, B)
    spec = net.spec()
    for sub in [spec] + spec["sublayers"] + [s.get("sublayer", {}) for s in spec["sublayers"]]:
        if not bias and "b" in sub:
            sub["b"] = None                          # the oracle returns gradients for the parameters that exist
    want_loss, want_acc, want = ot.loss_and_grads(spec, x, labels, weights, min_prob, l2, drop)
    step = train.TrainingStep(net, min_prob=min_prob, l2=l2, drop=drop)
    loss, acc = step.forward_backward(x, labels, weights)
    assert loss == pytest.approx(want_loss, rel=2e-5)
    assert acc == pytest.approx(want_acc, abs=1e-6)
    _assert_grads_close(step.gradients(), want)


@pytest.mark.parametrize("model,T,B", [("baseline_raw_gru", 120, 3), ("bigger_raw_gru", 90, 2), ("raw_1.00_rGr", 80, 3),
                                       ("baseline_gru", 45, 3), ("tiny_gru", 30, 5), ("baseline_lstm", 40, 3)])
def test_birnn_feedforward_models_vs_oracle(model, T, B):
    """models/baseline_raw_gru.py and bigger_raw_gru.py: convolution, then birnn (Parallel of a Gru and a reversed Gru,
    their outputs strided slices of one tensor) and FeedForward layers alternating; raw_1.00_rGr.py: 110- and 142-wide Gru
    layers, run zero-padded to 112 / 144; baseline_gru.py / tiny_gru.py: event features through a Window, Gru layers with 12
    inputs and (tiny) 4 neurons, zero-padded; baseline_lstm.py: birnn of peephole Lstm cells.  Gradients of every parameter."""
    need_gpu()
    from oracle import oracle_train as ot
    from sloika_amd import models, train
    net = models.randomise_zero_layers(models.build_model(model, klen=3, sd=0.5, seed=5))
    rs = np.random.RandomState(T)
    x, labels, weights = _batch(rs, net, T, B, nfeat=net.insize)
    want_loss, want_acc, want = ot.loss_and_grads(net.spec(), x, labels, weights, 1e-5, 0.001, 2)
    step = train.TrainingStep(net, min_prob=1e-5, l2=0.001, drop=2)
    loss, acc = step.forward_backward(x, labels, weights)
    assert loss == pytest.approx(want_loss, rel=2e-5) and acc == pytest.approx(want_acc, abs=1e-6)
    _assert_grads_close(step.gradients(), want)


def test_nested_serial_in_parallel_vs_oracle():
    """A Parallel whose branches are Serials (not plain layers): the generic path (per-branch forward, concatenate)."""
    need_gpu()
    from oracle import oracle_train as ot
    from sloika_amd import activation, layers, train
    rs = np.random.RandomState(8)
    init = lambda shape: (rs.normal(size=shape) * 0.5).astype(np.float32)
    n = 16
    branch = lambda: layers.Serial([layers.FeedForward(n, n, init=init, has_bias=True, fun=activation.relu),
                                    layers.Reverse(layers.Gru(n, n, init=init, has_bias=True))])
    net = layers.Serial([layers.Convolution(1, n, 5, 2, init=init, has_bias=True, fun=activation.tanh),
                         layers.Parallel([branch(), layers.Gru(n, 32, init=init, has_bias=True)]),
                         layers.FeedForward(n + 32, n, init=init, has_bias=False, fun=activation.sigmoid),
                         layers.Softmax(n, 7, init=init, has_bias=True)])
    x, labels, weights = _batch(rs, net, 50, 3)
    spec = net.spec()
    spec["sublayers"][2]["b"] = None
    want_loss, want_acc, want = ot.loss_and_grads(spec, x, labels, weights, 0.0, 0.0, 0)
    step = train.TrainingStep(net)
    loss, acc = step.forward_backward(x, labels, weights)
    assert loss == pytest.approx(want_loss, rel=2e-5) and acc == pytest.approx(want_acc, abs=1e-6)
    _assert_grads_close(step.gradients(), want)


@pytest.mark.parametrize("n,bias,peep,T,B", [(32, True, True, 33, 4), (16, False, False, 20, 2), (96, True, True, 25, 3),
                                            (24, True, True, 21, 3), (7, False, True, 12, 2), (80, True, False, 17, 2), (100, True, True, 9, 2)])
def test_lstm_stack_vs_oracle(n, bias, peep, T, B):
    """Lstm layers in both directions, with and without biases / peepholes, outside a Parallel (dense outputs); widths
    without a reverse-scan instantiation (24, 7, 80, 100) run zero-padded to the next one that has (32, 16, 96, 128)."""
    need_gpu()
    from oracle import oracle_train as ot
    from sloika_amd import layers, train
    rs = np.random.RandomState(n + T)
    init = lambda shape: (rs.normal(size=shape) * 0.5).astype(np.float32)
    net = layers.Serial([layers.Lstm(8, n, init=init, has_bias=bias, has_peep=peep),
                         layers.Reverse(layers.Lstm(n, n, init=init, has_bias=bias, has_peep=peep)),
                         layers.Softmax(n, 11, init=init, has_bias=True)])
    x, labels, weights = _batch(rs, net, T, B, nfeat=8)
    spec = net.spec()
    for sub in (spec["sublayers"][0], spec["sublayers"][1]["sublayer"]):
        if not bias:
            sub["b"] = None
        if not peep:
            sub["p"] = None
    want_loss, want_acc, want = ot.loss_and_grads(spec, x, labels, weights, 1e-6, 0.0, 1)
    step = train.TrainingStep(net, min_prob=1e-6, drop=1)
    loss, acc = step.forward_backward(x, labels, weights)
    assert loss == pytest.approx(want_loss, rel=2e-5) and acc == pytest.approx(want_acc, abs=1e-6)
    _assert_grads_close(step.gradients(), want)


def test_gru_only_network_and_device_inputs():
    """No convolution in front (the first Gru needs no dL/dx), inputs already on the device."""
    torch = need_gpu()
    from oracle import oracle_train as ot
    from sloika_amd import train
    rs = np.random.RandomState(3)
    net = _build(rs, n=16, nstate=9, nlayer=2, conv=False)
    x, labels, weights = _batch(rs, net, 30, 4, nfeat=16)
    want_loss, want_acc, want = ot.loss_and_grads(net.spec(), x, labels, weights, 0.0, 0.0, 0)
    step = train.TrainingStep(net)
    loss, acc = step.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda(),
                                      torch.from_numpy(weights).cuda())
    assert loss == pytest.approx(want_loss, rel=2e-5) and acc == pytest.approx(want_acc, abs=1e-6)
    _assert_grads_close(step.gradients(), want)


@pytest.mark.parametrize("entry", ["slk_gemm_tn_bf16x6_f32", "slk_gemm_tn_f32"])
@pytest.mark.parametrize("M,N1,N2", [(1, 1, 1), (37, 5, 3), (2049, 96, 11), (5000, 1025, 96), (4111, 288, 100), (300, 33, 1)])
def test_gemm_tn_vs_numpy(M, N1, N2, entry):
    """C = A^T B with operands inside wider rows (lda, ldb > N) and a padded result; the fp32-MFMA form and the six-term bf16 form."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    tn = getattr(L, entry)
    rs = np.random.RandomState(M)
    lda, ldb, ldc = N1 + 3, N2 + 5, N2 + 2
    A = rs.normal(size=(M, lda)).astype(np.float32)
    Bm = rs.normal(size=(M, ldb)).astype(np.float32)
    dA, dB = dev(A), dev(Bm)
    C = torch.full((N1, ldc), -7.0, dtype=torch.float32, device="cuda")
    nbytes = L.slk_gemm_tn_workspace_bytes(M, N1, N2)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    cs = torch.full((N1 + 1,), -7.0, dtype=torch.float32, device="cuda")
    assert tn(dA.data_ptr(), lda, dB.data_ptr(), ldb, C.data_ptr(), ldc, M, N1, N2, cs.data_ptr(), ws.data_ptr(),
                             nbytes, stream()) == 0
    want = A[:, :N1].astype(np.float64).T @ Bm[:, :N2].astype(np.float64)
    got = C.cpu().numpy()
    np.testing.assert_allclose(got[:, :N2], want, rtol=1e-5, atol=1e-5 * np.sqrt(M))
    assert (got[:, N2:] == -7.0).all()
    np.testing.assert_allclose(cs.cpu().numpy()[:N1], A[:, :N1].astype(np.float64).sum(0), rtol=1e-5, atol=1e-5 * np.sqrt(M))
    assert float(cs[N1]) == -7.0
    C.fill_(-7.0)
    assert tn(dA.data_ptr(), lda, dB.data_ptr(), ldb, C.data_ptr(), ldc, M, N1, N2, None, ws.data_ptr(), nbytes,
                             stream()) == 0
    np.testing.assert_allclose(C.cpu().numpy()[:, :N2], want, rtol=1e-5, atol=1e-5 * np.sqrt(M))
    assert tn(dA.data_ptr(), lda, dB.data_ptr(), ldb, C.data_ptr(), ldc, M, N1, N2, None, ws.data_ptr(),
                             nbytes - 1, stream()) == _lib.SLK_ERR_WORKSPACE



@pytest.mark.parametrize("amag,bmag", [(1e-30, 1e3), (1e-12, 1e-12), (1e20, 1e-25), (3e4, 7e4)])
def test_gemm_tn_bf16x6_any_magnitude(amag, bmag):
    """bf16 has float32's exponent range: gradients of 1e-30 and activations of 1e+4 need no scaling, and the six-term
    product keeps float32-grade accuracy relative to the size of the sum's terms."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(11)
    M, N1, N2 = 3000, 96, 40
    A = (rs.normal(size=(M, N1)) * amag * np.exp(rs.normal(size=(1, N1)) * 3)).astype(np.float32)     # columns of different sizes
    Bm = (rs.normal(size=(M, N2)) * bmag).astype(np.float32)
    C = torch.empty((N1, N2), dtype=torch.float32, device="cuda")
    nbytes = L.slk_gemm_tn_workspace_bytes(M, N1, N2)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    assert L.slk_gemm_tn_bf16x6_f32(dev(A).data_ptr(), N1, dev(Bm).data_ptr(), N2, C.data_ptr(), N2, M, N1, N2, None, ws.data_ptr(),
                                    nbytes, stream()) == 0
    want = A.astype(np.float64).T @ Bm.astype(np.float64)
    scale = np.sqrt((A.astype(np.float64) ** 2).sum(0))[:, None] * np.sqrt((Bm.astype(np.float64) ** 2).sum(0))[None, :]
    err = np.abs(C.cpu().numpy() - want) / scale
    assert err.max() < 2e-6, err.max()

@pytest.mark.parametrize("optimiser", ["adam", "sgd"])
def test_optimiser_kernel_vs_oracle(optimiser):
    """The same gradients through the update kernel and through the float32 restatement of updates.py:36-89 (adam) /
    :9-33 (sgd), five steps with a decaying rate; clip and l2 active."""
    torch = need_gpu()
    from oracle import oracle_train as ot
    from sloika_amd import train
    rs = np.random.RandomState(11)
    net = _build(rs, n=16, nstate=9, nlayer=1)
    l2 = 0.02
    step = train.TrainingStep(net, l2=l2, optimiser=optimiser, momentum=0.8)
    params = [p.get_value() for p in net.params()]
    opt = ot.Adamski(params)
    vel = [np.zeros_like(p) for p in params]
    for it in range(5):
        grads = [(rs.normal(size=p.shape) * (10.0 if it == 2 else 1.0)).astype(np.float32) for p in params]
        rate = 1e-2 / (1.0 + it)
        step.grad.copy_(torch.from_numpy(np.concatenate([g.reshape(-1) for g in grads])))
        step.update(rate)
        full = [g + np.float32(2 * l2) * p for g, p in zip(grads, params)]          # th.grad of loss incl. the penalty
        if optimiser == "adam":
            params = opt.step(params, full, rate)
        else:
            for k in range(len(params)):
                vel[k] = np.float32(0.8) * vel[k] - np.float32(rate) * np.clip(full[k], -5, 5)
                params[k] = params[k] + vel[k]
        step.sync_host()
        for p, want in zip(net.params(), params):
            np.testing.assert_allclose(p.get_value(), want, rtol=1e-5, atol=1e-6)


def test_training_reduces_loss_and_model_pickles(tmp_path):
    """fg(x, labels, weights, rate) as the reference's loop calls it (train_network.py:308): a learnable toy task (the
    label is a function of the local signal level) gets better, the inference path sees the updated weights, and the
    pickled checkpoint reloads to the same posteriors."""
    torch = need_gpu()
    from sloika_amd import helpers, train
    rs = np.random.RandomState(2)
    net = _build(rs, n=32, nstate=5, winlen=5, stride=2, nlayer=2, scale=0.3)
    fg = train.wrap_network(net, min_prob=1e-30, drop=2)
    T, B = 80, 16

    def batch():
        level = rs.randint(0, 5, size=(T // 2, B))
        x = np.repeat(level, 2, axis=0).astype(np.float32)[:, :, None] - 2.0 + 0.1 * rs.normal(size=(T, B, 1)).astype(np.float32)
        return x, level.astype(np.int32), np.ones((T // 2, B), dtype=np.float32)

    first = last = None
    for it in range(150):
        loss, acc = fg(*batch(), 3e-3)
        assert np.isfinite(loss)
        first = loss if first is None else first
        last = (loss, acc)
    assert last[0] < 0.5 * first and last[1] > 0.8, (first, last)
    x, labels, _ = batch()
    post = net.run(torch.from_numpy(x).cuda()).cpu().numpy()
    assert (post.argmax(2) == labels)[2:-2].mean() > 0.8
    # like a Theano shared variable under `updates`, a parameter shows what the optimiser wrote: get_value(), plain
    # pickling and set_value() all go through the step's device buffer
    import pickle
    w = net.layers[-1].W
    assert np.array_equal(w.get_value(), w.dev().cpu().numpy()) and np.abs(w.get_value()).max() > 0
    clone = pickle.loads(pickle.dumps(net))
    np.testing.assert_allclose(clone.run(torch.from_numpy(x).cuda()).cpu().numpy(), post, atol=1e-6)
    keep = w.get_value()
    w.set_value(np.zeros_like(keep))
    assert float(fg.flat[fg.offsets[-3]:fg.offsets[-2]].abs().max()) == 0.0 or float(w.dev().abs().max()) == 0.0
    w.set_value(keep)
    np.testing.assert_allclose(net.run(torch.from_numpy(x).cuda()).cpu().numpy(), post, atol=1e-6)
    path = train.save_model(net, str(tmp_path), index=1, step=fg)
    assert path.endswith("model_checkpoint_00001.pkl")
    again = helpers.load_model(path)
    np.testing.assert_allclose(again.run(torch.from_numpy(x).cuda()).cpu().numpy(), post, atol=1e-6)


def test_argument_validation():
    need_gpu()
    from sloika_amd import train
    rs = np.random.RandomState(1)
    net = _build(rs, n=16, nstate=5, nlayer=1)
    step = train.TrainingStep(net, drop=2)
    x, labels, weights = _batch(rs, net, 20, 2)
    with pytest.raises(ValueError):
        step.forward_backward(x, labels[:-1], weights[:-1])
    with pytest.raises(ValueError):
        step.forward_backward(x, labels + 5, weights)
    with pytest.raises(ValueError):
        train.TrainingStep(net, drop=5).forward_backward(x, labels, weights)


@pytest.mark.parametrize("n,reverse", [(96, 0), (32, 1), (144, 1)])
def test_gru_backward_kernels_agree(n, reverse):
    """The reverse scan has two kernels: operands through an LDS-DMA loader wave (16-byte aligned rows) and a plain one
    (any alignment).  Same inputs, dy once aligned and once shifted by one float: the same pre-activation gradients, and
    both equal the float64 recursion of oracle_train._backward's GRU step (the candidate c is handed over implicitly,
    through the layer output h_t = z h + (1-z) c)."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(n)
    T, B, I = 37, 5, 16
    M = T * B
    dy = rs.normal(size=(M, n)).astype(np.float32)
    xh = rs.normal(size=(M, I + n)).astype(np.float32)
    zr = rs.uniform(0.05, 0.95, size=(M, 2 * n)).astype(np.float32)
    c = rs.uniform(-0.95, 0.95, size=(M, n)).astype(np.float32)
    sW = (rs.normal(size=(2 * n, n)) / np.sqrt(n)).astype(np.float32)
    sW2 = (rs.normal(size=(n, n)) / np.sqrt(n)).astype(np.float32)
    # float64 recursion
    want = np.zeros((M, 3 * n))
    carry = np.zeros((B, n))
    for s in range(T - 1, -1, -1):
        t = T - 1 - s if reverse else s
        rows = slice(t * B, (t + 1) * B)
        g = dy[rows] + carry
        z, r, h, cc = zr[rows, :n].astype(np.float64), zr[rows, n:].astype(np.float64), xh[rows, I:].astype(np.float64), c[rows].astype(np.float64)
        dac = g * (1 - z) * (1 - cc * cc)
        daz = g * (h - cc) * z * (1 - z)
        drh = dac @ sW2
        dar = drh * h * r * (1 - r)
        carry = g * z + drh * r + np.concatenate([daz, dar], 1) @ sW
        want[rows] = np.concatenate([daz, dar, dac], 1)
    # the kernels take the layer's forward output h_t = z h + (1-z) c and recover the candidate from it
    hout = (zr[:, :n] * xh[:, I:] + (1.0 - zr[:, :n]) * c).astype(np.float32)
    d = {k: dev(v) for k, v in dict(xh=xh, zr=zr, hout=hout, sW=sW, sW2=sW2).items()}
    outs = []
    for shift in (0, 1):
        buf = torch.zeros(M * n + 4, dtype=torch.float32, device="cuda")
        buf[shift:shift + M * n] = dev(dy).reshape(-1)
        da = torch.empty((M, 3 * n), dtype=torch.float32, device="cuda")
        rh = torch.empty((M, n), dtype=torch.float32, device="cuda")
        rc = L.slk_gru_backward_f32(buf.data_ptr() + 4 * shift, n, d["xh"].data_ptr() + 4 * I, I + n, d["zr"].data_ptr(),
                                    d["hout"].data_ptr(), n, d["sW"].data_ptr(), d["sW2"].data_ptr(), da.data_ptr(),
                                    rh.data_ptr(), T, B, n, reverse, 1, 2, stream())
        assert rc == 0
        outs.append(da.cpu().numpy())
        np.testing.assert_array_equal(rh.cpu().numpy(), zr[:, n:] * xh[:, I:])
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-4, atol=1e-5 * np.abs(want).max())     # different summation orders
    np.testing.assert_allclose(outs[0], want, rtol=1e-4, atol=1e-4 * np.abs(want).max())
    assert L.slk_gru_backward_f32(d["xh"].data_ptr(), 160, d["xh"].data_ptr(), 160, d["zr"].data_ptr(), d["hout"].data_ptr(), 160,
                                  d["sW"].data_ptr(), d["sW2"].data_ptr(), da.data_ptr(), rh.data_ptr(), T, B, 40, reverse, 1,
                                  2, stream()) == _lib.SLK_ERR_UNSUPPORTED


def test_train_loop_end_to_end(tmp_path):
    """train_network.py:180-330 as a function: chunk file -> sampler -> fg -> log, checkpoints, final model.  The toy task
    (label = quantised local signal level, half of the positions blank) must be learnt."""
    torch = need_gpu()
    from sloika_amd import helpers, train
    rs = np.random.RandomState(3)
    n, clen, stride = 64, 200, 2
    level = rs.randint(1, 5, size=(n, clen // stride))
    chunks = (np.repeat(level, stride, axis=1).astype(np.float32) - 2.5 + 0.1 * rs.normal(size=(n, clen)))[:, :, None]
    labels = level.astype(np.int32)
    labels[:, 1::2] = 0                                                   # blanks
    chunks[:, 2::4, 0] += 3.0                                             # ... which the signal marks
    chunks[:, 3::4, 0] += 3.0
    path = os.path.join(str(tmp_path), "chunks.npz")
    np.savez(path, chunks=chunks.astype(np.float32), labels=labels, bad=np.zeros_like(labels, dtype='i1'),
             weights=np.ones(n, dtype=np.float32), kmer=np.int64(1), alphabet=np.bytes_(b"ACGT"))
    net = _build(np.random.RandomState(4), n=32, nstate=5, winlen=5, stride=stride, nlayer=2, scale=0.3)
    out = os.path.join(str(tmp_path), "run")
    fg = train.train_loop(net, train.load_chunk_file(path), out, niteration=200, batch_size=32, drop=4, adam=(4e-3, 0.9, 0.999),
                          save_every=100, seed=9, quiet=True)
    files = sorted(os.listdir(out))
    assert files == ["model.log", "model_checkpoint_00000.pkl", "model_checkpoint_00001.pkl", "model_checkpoint_00002.pkl",
                     "model_final.pkl"]
    log = open(os.path.join(out, "model.log")).read()
    assert log.count("C") >= 2 and log.count(".") >= 198 and "kev/s" in log
    lines = [l for l in log.splitlines() if "kev/s" in l]
    pct = lambda line: float([tok for tok in line.split() if tok.endswith("%")][0].rstrip("%"))
    first_acc, last_acc = pct(lines[0]), pct(lines[-1])
    assert last_acc > 90.0 and last_acc > first_acc, (first_acc, last_acc)
    final = helpers.load_model(os.path.join(out, "model_final.pkl"))
    x = torch.from_numpy(np.ascontiguousarray(chunks[:8].transpose(1, 0, 2)).astype(np.float32)).cuda()
    post = final.run(x).cpu().numpy()
    assert (post.argmax(2) == labels[:8].T)[4:-4].mean() > 0.9
    initial = helpers.load_model(os.path.join(out, "model_checkpoint_00000.pkl"))
    assert (initial.run(x).cpu().numpy().argmax(2) == labels[:8].T)[4:-4].mean() < 0.6


def test_full_size_gradient_is_mean_over_batch_halves():
    """BASELINE.json's shape (raw_0.98_rgrgr, 4000-sample chunks) is too large for the float64 oracle; the size-independent
    property: the loss is a mean over chunks, so the gradient of a batch equals the mean of the gradients of its halves
    (this is also exactly what the data-parallel all-reduce relies on).  512 chunks = 409600 rows per contraction."""
    torch = need_gpu()
    from sloika_amd import models, train
    net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=3))
    step = train.TrainingStep(net, min_prob=1e-30, drop=20)
    rs = np.random.RandomState(0)
    B, T = 512, 4000
    x = torch.from_numpy(rs.normal(size=(T, B, 1)).astype(np.float32)).cuda()
    To = net.layers[0].out_len(T)
    labels = torch.from_numpy(rs.randint(0, net.size, size=(To, B)).astype(np.int32)).cuda()
    weights = torch.from_numpy(rs.uniform(0.5, 1.5, size=(To, B)).astype(np.float32)).cuda()
    loss, acc = step.forward_backward(x, labels, weights)
    whole = step.gradients()
    halves, losses = [], []
    for sl in (slice(0, B // 2), slice(B // 2, B)):
        l, _ = step.forward_backward(x[:, sl].contiguous(), labels[:, sl].contiguous(), weights[:, sl].contiguous())
        halves.append(step.gradients())
        losses.append(l)
    assert loss == pytest.approx(0.5 * (losses[0] + losses[1]), rel=1e-5)
    assert abs(loss - np.log(1025.0)) < 0.5 and np.isfinite(loss)                 # random weights: near the uniform posterior
    for w, a, b in zip(whole, halves[0], halves[1]):
        scale = max(float(np.abs(w).max()), 1e-12)
        np.testing.assert_allclose(w / scale, 0.5 * (a + b) / scale, atol=2e-4)
        assert np.isfinite(w).all() and np.abs(w).max() > 0


def test_training_edge_cases():
    """Smallest shapes (one chunk, three output steps), zero label weights (only the l2 term is left: gradient = 2 l2 theta,
    loss = l2 |theta|^2), plain ADAM (mrate=None) and SGD taking a step without producing NaNs."""
    need_gpu()
    from oracle import oracle_train as ot
    from sloika_amd import train
    rs = np.random.RandomState(12)
    net = _build(rs, n=16, nstate=5, winlen=3, stride=2, nlayer=2)
    x, labels, weights = _batch(rs, net, 6, 1)                       # T' = 3, B = 1
    want_loss, want_acc, want = ot.loss_and_grads(net.spec(), x, labels, weights, 1e-4, 0.0, 1)
    step = train.TrainingStep(net, min_prob=1e-4, drop=1)
    loss, acc = step.forward_backward(x, labels, weights)
    assert loss == pytest.approx(want_loss, rel=2e-5) and acc == pytest.approx(want_acc, abs=1e-6)
    _assert_grads_close(step.gradients(), want)
    # zero weights: the data term vanishes
    l2 = 0.05
    step = train.TrainingStep(net, l2=l2, drop=1)
    loss, _ = step.forward_backward(x, labels, np.zeros_like(weights))
    params = [p.get_value() for p in net.params()]
    assert loss == pytest.approx(l2 * sum(float(np.sum(np.square(p.astype(np.float64)))) for p in params), rel=1e-5)
    for g, p in zip(step.gradients(), params):
        np.testing.assert_allclose(g, 2 * l2 * p, rtol=1e-5, atol=1e-7)
    # optimiser variants take finite steps
    for kw in (dict(mrate=None), dict(optimiser="sgd", momentum=0.5)):
        fg = train.TrainingStep(net, min_prob=1e-4, drop=1, **kw)
        before = [p.get_value() for p in net.params()]
        first, _ = fg(x, labels, weights, 1e-2)
        for _ in range(20):
            last, _ = fg(x, labels, weights, 1e-2)
        after = [p.get_value() for p in net.params()]
        assert np.isfinite(last) and last < first
        assert all(np.isfinite(a).all() for a in after) and any(np.abs(a - b).max() > 0 for a, b in zip(after, before))


@pytest.mark.parametrize("K,N,T,B,drop,min_prob", [
    (96, 1025, 83, 7, 2, 1e-30),       # the headline layer; 581 rows: ragged last workgroup, partial last column tile
    (64, 1025, 40, 16, 0, 1e-5),       # 640 rows: whole workgroups (the branch-free epilogue)
    (112, 260, 33, 5, 3, 0.0),
    (128, 17, 50, 4, 1, 1e-3),         # one partial tile only
    (96, 1024, 16, 8, 0, 1e-30),       # exactly sixteen full tiles, ld == N
])
def test_softmax_loss_gradient_two_passes_equal_in_place(K, N, T, B, drop, min_prob):
    """slk_linear_xent_grad_f16x3 (two passes over the products, no logits in memory) against the pair it replaces
    (slk_linear_rowstats_f16x3, then slk_softmax_xent_grad_f32 in place): the same bits, including the first-maximum rule of
    T.argmax on rows with ties and the zeroed padding columns."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(K + N)
    M, ld, kp = T * B, (N + 31) // 32 * 32, (K + 15) // 16 * 16
    x = rs.normal(size=(M, K)).astype(np.float32)
    W = (rs.normal(size=(N, K)) * 0.4).astype(np.float32)
    b = rs.normal(size=N).astype(np.float32)
    W[N - 3] = W[2]; b[N - 3] = b[2]                           # two columns with identical logits in every row ...
    labels = rs.randint(0, N, size=M).astype(np.int32)
    r0 = M // 2                                                # (a counted row: drop <= t < T - drop)
    x[r0] = 0.0; b[2] = b[N - 3] = b.max() + 1.0              # ... and they hold the maximum of many rows (always of row r0)
    labels[r0], labels[r0 + 1], labels[r0 + 2] = 2, N - 3, N - 1
    weights = rs.uniform(0.5, 1.5, size=M).astype(np.float32)
    xd, Wd, bd, ld_, wd = dev(x), dev(W), dev(b), dev(labels), dev(weights)
    hi = torch.empty((N, kp), dtype=torch.float16, device="cuda"); lo = torch.empty_like(hi)
    inv = torch.empty(N, dtype=torch.float32, device="cuda")
    _lib.check(L.slk_split_f16x2_f32(Wd.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), stream()), "split")
    # the pair
    logits = torch.full((M, ld), 7.0, dtype=torch.float32, device="cuda")
    stats = torch.empty((M, 2), dtype=torch.float32, device="cuda")
    rows_a = torch.empty((2, M), dtype=torch.float32, device="cuda")
    _lib.check(L.slk_linear_rowstats_f16x3(xd.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), bd.data_ptr(),
                                           logits.data_ptr(), ld, M, K, N, stats.data_ptr(), stream()), "rowstats")
    _lib.check(L.slk_softmax_xent_grad_f32(logits.data_ptr(), ld, stats.data_ptr(), ld_.data_ptr(), wd.data_ptr(), T, B, N, drop,
                                           min_prob, rows_a[0].data_ptr(), rows_a[1].data_ptr(), stream()), "xent")
    # the two passes
    grad = torch.full((M, ld), 7.0, dtype=torch.float32, device="cuda")
    rows_b = torch.empty((2, M), dtype=torch.float32, device="cuda")
    xrow = torch.empty((M, 4), dtype=torch.float32, device="cuda")
    _lib.check(L.slk_linear_xent_grad_f16x3(xd.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), bd.data_ptr(),
                                            grad.data_ptr(), ld, K, N, ld_.data_ptr(), wd.data_ptr(), T, B, drop, min_prob,
                                            rows_b[0].data_ptr(), rows_b[1].data_ptr(), xrow.data_ptr(), stream()), "two passes")
    torch.cuda.synchronize()
    assert torch.equal(rows_a, rows_b)
    assert torch.equal(grad, logits)
    assert float(rows_b[1][r0]) > 0.0                          # the FIRST of the tied maxima is the label: counted correct
    assert not bool((grad[:, N:] != 0).any())
    # K outside the instantiated widths is refused, not computed some other way
    rc = L.slk_linear_xent_grad_f16x3(xd.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), bd.data_ptr(), grad.data_ptr(), ld,
                                      32, N, ld_.data_ptr(), wd.data_ptr(), T, B, drop, min_prob, rows_b[0].data_ptr(),
                                      rows_b[1].data_ptr(), xrow.data_ptr(), stream())
    assert rc == _lib.SLK_ERR_UNSUPPORTED


def test_training_step_two_pass_softmax_equals_in_place():
    """TrainingStep.forward_backward with and without train.XENT_TWO_PASS: same loss, accuracy and gradients bit for bit."""
    need_gpu()
    from sloika_amd import train
    rs = np.random.RandomState(11)
    net = _build(rs, n=96, nstate=1025, stride=5, winlen=11, nlayer=2)
    x, labels, weights = _batch(rs, net, 200, 6)
    step = train.TrainingStep(net, min_prob=1e-30, drop=2)
    res = []
    for flag in (True, False):
        train.XENT_TWO_PASS = flag
        try:
            la = step.forward_backward(x, labels, weights)
        finally:
            train.XENT_TWO_PASS = True
        res.append((la, [g.copy() for g in step.gradients()]))
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1], res[1][1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("M,n,i_sz", [(4111, 96, 96), (900, 64, 33), (2049, 128, 40)])
def test_gemm_tn_multi_equals_single_launches(M, n, i_sz):
    """slk_gemm_tn_multi_bf16x6_f32 (the three weight gradients of a Gru layer in one launch, sharing dL/d(pre-activation)) gives what three
    slk_gemm_tn_bf16x6_f32 launches give (to the rounding of float32 sums grouped differently) and leaves the padding of its outputs alone."""
    import ctypes
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(M + n)
    da = dev((rs.normal(size=(M, 3 * n)) * 1e-3).astype(np.float32))
    xs = [dev(rs.normal(size=(M, w + 2)).astype(np.float32)) for w in (i_sz, n, n)]
    shapes = [(3 * n, i_sz, 0), (2 * n, n, 0), (n, n, 2 * n)]            # (N1, N2, first column of da)
    single, multi, cs_single, cs_multi = [], [], torch.full((3 * n,), -7.0, device="cuda"), torch.full((3 * n,), -7.0, device="cuda")
    for (n1, n2, c0), xb in zip(shapes, xs):
        C = torch.full((n1, n2 + 1), -7.0, dtype=torch.float32, device="cuda")
        nbytes = L.slk_gemm_tn_workspace_bytes(M, n1, n2)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        assert L.slk_gemm_tn_bf16x6_f32(da.data_ptr() + 4 * c0, 3 * n, xb.data_ptr(), n2 + 2, C.data_ptr(), n2 + 1, M, n1, n2,
                                        cs_single.data_ptr() if c0 == 0 and n1 == 3 * n else None, ws.data_ptr(), nbytes, stream()) == 0
        single.append(C)
        multi.append(torch.full((n1, n2 + 1), -7.0, dtype=torch.float32, device="cuda"))
    vps, longs, ints = ctypes.c_void_p * 3, ctypes.c_long * 3, ctypes.c_int * 3
    n1s, n2s = ints(*[s[0] for s in shapes]), ints(*[s[1] for s in shapes])
    nbytes = L.slk_gemm_tn_multi_workspace_bytes(M, 3, n1s, n2s)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    args = (3, vps(*[da.data_ptr() + 4 * s[2] for s in shapes]), longs(3 * n, 3 * n, 3 * n), vps(*[x.data_ptr() for x in xs]),
            longs(*[s[1] + 2 for s in shapes]), vps(*[c.data_ptr() for c in multi]), longs(*[s[1] + 1 for s in shapes]), M, n1s, n2s,
            vps(cs_multi.data_ptr(), None, None), ws.data_ptr())
    assert L.slk_gemm_tn_multi_bf16x6_f32(*args, nbytes, stream()) == 0
    torch.cuda.synchronize()
    for a, b, (n1, n2, c0) in zip(single, multi, shapes):           # (the slices of the rows differ: float32 sums in another grouping)
        scale = float(a[:, :n2].abs().max())
        assert float((a[:, :n2] - b[:, :n2]).abs().max()) <= 2e-6 * scale
        assert bool((b[:, n2:] == -7.0).all())
    assert float((cs_single - cs_multi).abs().max()) <= 2e-6 * float(cs_single.abs().max())
    assert L.slk_gemm_tn_multi_bf16x6_f32(*args, nbytes - 1, stream()) == _lib.SLK_ERR_WORKSPACE
    assert L.slk_gemm_tn_multi_bf16x6_f32(5, *args[1:], nbytes, stream()) == _lib.SLK_ERR_INVALID_ARG


def test_out_of_range_labels_raise_and_leave_the_parameters_alone():
    """The two-pass softmax path checks the labels on the device, beside the step (no host synchronisation in the middle of it): the
    step still raises, and it raises before the optimiser has touched a parameter."""
    need_gpu()
    from sloika_amd import train
    rs = np.random.RandomState(3)
    net = _build(rs, n=64, nstate=260, nlayer=1)
    step = train.TrainingStep(net, drop=1)
    x, labels, weights = _batch(rs, net, 40, 3)
    before = [p.get_value().copy() for p in net.params()]
    bad = labels.copy()
    bad[7, 1] = 260
    with pytest.raises(ValueError):
        step(x, bad, weights, 1e-3)
    bad[7, 1] = -1
    with pytest.raises(ValueError):
        step.forward_backward(x, bad, weights)
    for a, p in zip(before, net.params()):
        np.testing.assert_array_equal(a, p.get_value())
    loss, acc = step(x, labels, weights, 1e-3)                     # ... and the step object is still usable
    assert np.isfinite(loss) and any((a != p.get_value()).any() for a, p in zip(before, net.params()))


@pytest.mark.parametrize("n", [1, 255, 819200, 70001])
def test_reduce_rows_sum(n):
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(n % 97)
    x = (rs.normal(size=(3, n)) * 10.0 ** rs.uniform(-3, 3, size=(3, 1))).astype(np.float32)
    xd = dev(x)
    out = torch.zeros(3, dtype=torch.float64, device="cuda")
    scratch = torch.empty(3 * 256, dtype=torch.float64, device="cuda")
    assert L.slk_reduce_rows_sum_f32(xd.data_ptr(), 3, n, out.data_ptr(), scratch.data_ptr(), stream()) == 0
    want = x.astype(np.float64).sum(axis=1)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-12, atol=1e-9 * np.abs(x).max())
    again = torch.zeros(3, dtype=torch.float64, device="cuda")
    assert L.slk_reduce_rows_sum_f32(xd.data_ptr(), 3, n, again.data_ptr(), scratch.data_ptr(), stream()) == 0
    assert torch.equal(out, again)                                 # fixed order: the same bits every time


@pytest.mark.parametrize("act", ["tanh", "elu", "relu", "sigmoid"])
def test_dx_product_with_the_activation_derivative_of_the_layer_below(act):
    """slk_gemm_dact_bf16x6 = slk_gemm_bias_act_bf16x6 followed by slk_act_backward_f32, bit for bit; and the training step that uses it
    (a Gru layer over a Convolution) gives the gradients of the step that does not (checked against the oracle elsewhere)."""
    torch = need_gpu()
    from sloika_amd import _lib, activation
    L = _lib.lib()
    rs = np.random.RandomState(17)
    M, K, N = 1000, 288, 96
    x = dev((rs.normal(size=(M, K)) * 1e-3).astype(np.float32))
    W = dev((rs.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32))
    yref = dev(np.tanh(rs.normal(size=(M, N + 4))).astype(np.float32))           # rows ldyref = N + 4 apart
    packed = torch.empty(L.slk_pack_bf16x3_bytes(N, K), dtype=torch.uint8, device="cuda")
    assert L.slk_pack_bf16x3_f32(W.data_ptr(), N, K, packed.data_ptr(), stream()) == 0
    aid = activation.act_id(getattr(activation, act))
    fused = torch.empty((M, N), dtype=torch.float32, device="cuda")
    assert L.slk_gemm_dact_bf16x6(x.data_ptr(), K, packed.data_ptr(), yref.data_ptr(), N + 4, aid, fused.data_ptr(), N, M, K, N, stream()) == 0
    plain = torch.empty((M, N), dtype=torch.float32, device="cuda")
    assert L.slk_gemm_bias_act_bf16x6(x.data_ptr(), K, packed.data_ptr(), None, plain.data_ptr(), N, M, K, N, 0, stream()) == 0
    yc = yref[:, :N].contiguous()
    two = torch.empty_like(plain)
    assert L.slk_act_backward_f32(plain.data_ptr(), yc.data_ptr(), two.data_ptr(), plain.numel(), aid, stream()) == 0
    assert torch.equal(fused, two)
