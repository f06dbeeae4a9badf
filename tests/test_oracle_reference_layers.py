"""The oracles against numbers the REFERENCE'S OWN layer code produced (tests/golden/layers.npz): sloika/layers.py,
conv.py, activation.py, models/*.py, models/pretrained.pkl, basecall.raw_worker and bin/train_network.py:wrap_network +
sloika/updates.py:adam executed unmodified under the eager Theano stand-in of tests/golden/theano_standin (see
tests/golden/make_layer_goldens.py for what that does and does not pin).

  * numpy float64 oracle vs the float64 evaluation of the reference's code: <= 1e-10 -- the formulas are IDENTICAL
    (gate order, reshapes, padding, scan order, initial state), not merely close;
  * C float32 oracle (what the HIP kernels and bench.py's cpu_baseline are checked against): <= 2e-5.
"""
import os

import numpy as np
import pytest

from tests import ref_layers as rl
from tests.ref_layers import lc
from oracle import oracle_np, oracle_train as ot

META = rl.meta()


@pytest.mark.parametrize("name", sorted(META["layers"]))
def test_layer_vs_reference_code(oracle, name):
    case = META["layers"][name]
    rl.check_inputs(case)
    want = rl.arrays()["layer/" + name]
    assert want.dtype == np.float64
    x = lc.expand(case["x"])
    y64 = oracle_np.run_network(lc.materialise(case["tree"], np.float64), x.astype(np.float64))
    assert y64.shape == want.shape
    np.testing.assert_allclose(y64, want, rtol=0, atol=1e-10)
    acts = {leaf.get(k) for leaf in lc.walk(case["tree"]) for k in ("activation", "gate")} - {None}
    if acts <= set(oracle.ACTIVATIONS):
        y32 = oracle.run_network(lc.materialise(case["tree"]), x)
        np.testing.assert_allclose(y32, want, rtol=0, atol=2e-5)


@pytest.mark.parametrize("name", sorted(META["models"]))
def test_model_factory_vs_reference_code(oracle, name):
    """models/<name>.py called unchanged by the generator; here: the same layer sequence through both oracles, and this
    package's own tabulation of the factory (sloika_amd/models.py) must have the same parameter list."""
    case = META["models"][name]
    rl.check_inputs(case)
    want = rl.arrays()["model/" + name]
    x = lc.expand(case["x"])
    y64 = oracle_np.run_network(lc.materialise(case["tree"], np.float64), x.astype(np.float64))
    assert y64.shape == want.shape and want.shape[2] == 1025
    np.testing.assert_allclose(y64, want, rtol=2e-6, atol=1e-9)            # stored rounded to float32
    y32 = oracle.run_network(lc.materialise(case["tree"]), x)
    scale = want.max(axis=2, keepdims=True)
    assert np.abs(y32 - want).max() < 2e-5 and (np.abs(y32 - want) / scale).max() < 2e-4
    from sloika_amd import models
    net = models.build_model(name, klen=5, sd=0.5, seed=1)
    assert [list(p.get_value().shape) for p in net.params()] == case["param_shapes"]
    assert net.insize == case["x"]["shape"][2] and net.size == 1025


@pytest.mark.parametrize("mode,winlen", [("same", 11), ("same", 4), ("half", 6), ("valid", 5), ("full", 5), ("same_left", 4),
                                         (2, 7), (0, 9)])
def test_calculate_padding_equals_reference(mode, winlen):
    from sloika_amd import conv
    seen = {(leaf["padding_mode"], leaf["winlen"]): tuple(leaf["padding"])
            for c in META["layers"].values() for leaf in lc.walk(c["tree"]) if leaf["type"] == "convolution"}
    assert tuple(conv.calculate_padding(mode, winlen)) == seen[(mode, winlen)]


def _signal_of_read(n, nsample):
    g = np.load(os.path.join(rl.GOLDEN, "reads.npz"))
    dig, off, rng, _ = g["meta_%d" % n]
    return ((g["adc_%d" % n].astype(np.float64) + off) * (rng / dig))[:nsample]


def test_pretrained_pickle_on_real_read(oracle):
    """models/pretrained.pkl unpickled INTO THE REFERENCE'S CLASSES and driven by the reference's basecall.raw_worker on
    the first 12000 samples of data/reads/read5.fast5; the oracle gets the same weights from pretrained_weights.npz."""
    p = META["pretrained"]
    A = rl.arrays()
    signal = _signal_of_read(p["read"], p["nsample"])
    signal = signal[p["trim"][0]: len(signal) - p["trim"][1]]                 # open_pore_fraction 0: trim_open_pore keeps all
    med = np.median(signal)
    inmat = ((signal - med) / (1.4826 * np.median(np.abs(signal - med)))).astype(np.float32)
    np.testing.assert_allclose(inmat[:64], A["pretrained/read5_inmat_head"], rtol=1e-6)
    w = np.load(os.path.join(rl.GOLDEN, "pretrained_weights.npz"))
    import json
    subs = []
    for d in json.loads(str(w["description_json"])):
        i = d["index"]
        if d["type"] == "Convolution":
            node = {"type": "convolution", "W": w["l%d_W" % i], "b": w["l%d_b" % i], "stride": d["stride"],
                    "padding": tuple(d["padding"]), "activation": d["fun"]}
        elif d["type"] == "Gru":
            node = {"type": "GRU", "iW": w["l%d_iW" % i], "sW": w["l%d_sW" % i], "sW2": w["l%d_sW2" % i], "b": w["l%d_b" % i],
                    "activation": d["fun"], "gate": d["gatefun"]}
        else:
            node = {"type": "softmax", "W": w["l%d_W" % i], "b": w["l%d_b" % i]}
        subs.append({"type": "reverse", "sublayer": node} if d["reverse"] else node)
    post = oracle.run_network({"type": "serial", "sublayers": subs}, inmat[:, None, None])
    assert post.shape[0] == p["nstep"] and p["skip0"]["nsamp"] == len(inmat)
    rows = A["pretrained/read5_post_rows"]
    np.testing.assert_allclose(post[::p["every"], 0, :], rows, rtol=0, atol=2e-5)
    for skip in (0.0, 5.0):
        score, path = oracle.viterbi(oracle.prepare_post(post, 1e-5), 5, skip_pen=skip)
        want = A["pretrained/read5_call_skip%g" % skip]
        assert score == pytest.approx(p["skip%g" % skip]["score"], rel=2e-5)
        # identical calls except where float32-vs-float64 posteriors flip a near-tie
        assert rl.int_edit_distance(np.asarray(path), want) <= 0.002 * len(want)


@pytest.mark.parametrize("name", sorted(META["train"]))
def test_training_step_vs_reference_code(name):
    """wrap_network's loss / accuracy, th.grad of it (autograd over the reference's own graph) and the parameters after
    `steps` ADAMski updates, against oracle_train (hand-derived reverse pass, float32 optimiser arithmetic)."""
    c = META["train"][name]
    rl.check_inputs(c)
    A = rl.arrays()
    labels, weights = A["train/%s/labels" % name], A["train/%s/weights" % name]
    x = lc.expand(c["x"])
    spec = lc.materialise(c["tree"], np.float64)
    loss, acc, grads = ot.loss_and_grads(spec, x.astype(np.float64), labels, weights, c["min_prob"], c["l2"], c["drop"])
    assert loss == pytest.approx(c["hist"][0][0], rel=1e-9) and acc == pytest.approx(c["hist"][0][1], abs=1e-12)
    for k, g in enumerate(grads):
        want = A["train/%s/grad%d" % (name, k)]
        assert g.shape == want.shape
        np.testing.assert_allclose(g, want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()))
    # optimiser: `steps` calls of fg(x, labels, weights, rate)   (train_network.py:308, updates.py:36-89)
    spec32 = lc.materialise(c["tree"])
    params = ot.params_of(spec32)
    opt = ot.Adamski(params, decay=tuple(c["adam"]))
    for step in range(c["steps"]):
        lo, ac, gr = ot.loss_and_grads(spec32, x, labels, weights, c["min_prob"], c["l2"], c["drop"])
        assert lo == pytest.approx(c["hist"][step][0], rel=2e-5) and ac == pytest.approx(c["hist"][step][1], abs=1e-6)
        for p_, new in zip(params, opt.step(params, gr, c["rate"])):
            p_[...] = new
    for k, p_ in enumerate(params):
        want = A["train/%s/param%d" % (name, k)]
        np.testing.assert_allclose(p_, want, rtol=0, atol=2e-6)
