"""The HIP path (through the C ABI) against numbers the REFERENCE'S OWN layer code produced: tests/golden/layers.npz, made
by tests/golden/make_layer_goldens.py from sloika/layers.py, conv.py, models/*.py, models/pretrained.pkl,
basecall.raw_worker and bin/train_network.py:wrap_network + updates.adam (see that script for what is pinned).

Tolerances: layer outputs (all in [-1, 1] or O(1)) 1e-4 absolute as north_star states, observed ~1e-6; posteriors of whole
models additionally relative to each row's largest posterior (2e-4), since 1025-way posteriors average 1e-3.
"""
import os

import numpy as np
import pytest

from tests import ref_layers as rl
from tests.ref_layers import lc
from tests.gpu_util import need_gpu

pytestmark = pytest.mark.gpu
META = rl.meta()
TOL = 1e-4


@pytest.mark.parametrize("name", sorted(META["layers"]))
def test_layer_vs_reference_code(name):
    need_gpu()
    case = META["layers"][name]
    rl.check_inputs(case)
    want = rl.arrays()["layer/" + name]
    y = rl.build_amd(case["tree"]).compile()(lc.expand(case["x"]))
    assert y.shape == want.shape and y.dtype == np.float32
    err = np.abs(y - want).max()
    assert err < (2e-5 if name.startswith(("conv", "window", "ff", "softmax")) else TOL), err


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("name", sorted(META["models"]))
def test_model_factory_vs_reference_code(name, exact, monkeypatch):
    """sloika_amd.models' tabulation of models/<name>.py, loaded with the weights the reference's factory output was given,
    against the posteriors the reference's network.compile() returned -- in the default arithmetic and in all-fp32 mode."""
    need_gpu()
    from sloika_amd import layers, models
    monkeypatch.setattr(layers, "SPLIT_F16", not exact)
    case = META["models"][name]
    rl.check_inputs(case)
    want = rl.arrays()["model/" + name]
    net = models.build_model(name, klen=5, sd=0.5, seed=1)
    arrays = lc.param_arrays(case["tree"])
    params = net.params()
    assert len(params) == len(arrays)
    for p, a in zip(params, arrays):
        assert tuple(p.get_value().shape) == a.shape
        p.set_value(a)
    y = net.compile()(lc.expand(case["x"]))
    assert y.shape == want.shape
    assert np.abs(y - want).max() < 2e-5
    assert (np.abs(y - want) / want.max(axis=2, keepdims=True)).max() < 2e-4


def _signal_of_read(n, nsample):
    g = np.load(os.path.join(rl.GOLDEN, "reads.npz"))
    dig, off, rng, _ = g["meta_%d" % n]
    return ((g["adc_%d" % n].astype(np.float64) + off) * (rng / dig))[:nsample]


@pytest.mark.parametrize("skip", [0.0, 5.0])
def test_pretrained_pickle_on_real_read(skip):
    """The reference's trained model on a real read, whole-read mode: posteriors and the basecall itself against what the
    reference's classes + basecall.raw_worker produced from models/pretrained.pkl."""
    need_gpu()
    from sloika_amd import basecall, bio, models
    p = META["pretrained"]
    A = rl.arrays()
    net = models.from_weights_npz(os.path.join(rl.GOLDEN, "pretrained_weights.npz"))
    seen = {}
    calc = net.compile()

    def calc_post(inmat):
        seen["post"] = calc(inmat)
        return seen["post"]
    signal = _signal_of_read(p["read"], p["nsample"])
    name, score, call, nsamp = basecall.raw_read_worker(calc_post, signal, trim=tuple(p["trim"]), kmer_len=5, skip=skip,
                                                        min_prob=p["min_prob"], name="read5")
    ref = p["skip%g" % skip]
    assert nsamp == ref["nsamp"]
    post = np.asarray(seen["post"].cpu() if hasattr(seen["post"], "cpu") else seen["post"])
    rows = A["pretrained/read5_post_rows"]
    assert post.shape[0] == p["nstep"]
    got = post[::p["every"], 0, :]
    assert np.abs(got - rows).max() < 5e-5
    assert float(score) == pytest.approx(ref["score"], rel=2e-5)
    want = A["pretrained/read5_call_skip%g" % skip]
    assert rl.int_edit_distance(np.asarray(call), want) <= 0.002 * len(want)
    kmers = bio.all_kmers(5)
    seq = bio.kmers_to_sequence([kmers[i] for i in call], always_move=True)
    assert rl.int_edit_distance(np.frombuffer(seq.encode(), np.uint8), np.frombuffer(ref["seq"].encode(), np.uint8)) \
        <= 0.002 * len(ref["seq"])


@pytest.mark.parametrize("name", sorted(META["train"]))
def test_training_step_vs_reference_code(name):
    """Loss, accuracy, every gradient and the parameters after `steps` ADAMski updates against the reference's
    wrap_network / updates.adam (gradient = automatic differentiation of the reference's own graph)."""
    need_gpu()
    from sloika_amd import train
    c = META["train"][name]
    rl.check_inputs(c)
    A = rl.arrays()
    labels, weights = A["train/%s/labels" % name], A["train/%s/weights" % name]
    x = lc.expand(c["x"])
    net = rl.build_amd(c["tree"])
    step = train.TrainingStep(net, min_prob=c["min_prob"], l2=c["l2"], drop=c["drop"], decay=tuple(c["adam"]))
    loss, acc = step.forward_backward(x, labels, weights)
    assert loss == pytest.approx(c["hist"][0][0], rel=2e-5) and acc == pytest.approx(c["hist"][0][1], abs=1e-6)
    for k, g in enumerate(step.gradients()):
        want = A["train/%s/grad%d" % (name, k)]
        assert g.shape == want.shape
        scale = max(float(np.abs(want).max()), 1e-6)
        np.testing.assert_allclose(g / scale, want / scale, atol=2e-4, err_msg="parameter %d" % k)
    step.update(c["rate"])
    for s in range(1, c["steps"]):
        loss, acc = step(x, labels, weights, c["rate"])
        assert loss == pytest.approx(c["hist"][s][0], rel=5e-5) and acc == pytest.approx(c["hist"][s][1], abs=1e-6)
    step.sync_host()
    for k, p_ in enumerate(net.params()):
        want = A["train/%s/param%d" % (name, k)]
        np.testing.assert_allclose(p_.get_value(), want, rtol=0, atol=1e-5, err_msg="parameter %d" % k)
