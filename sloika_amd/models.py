"""Architectures of the reference's shipped models, built through this package's layer API.

The reference keeps them as executable factories under models/*.py, which `helpers.load_factory` runs unchanged
when that checkout is at hand.  The GPU box only receives this repository, so the same layer sequences are
tabulated here (structure only -- sizes, activations, directions -- read off the cited lines):

    tiny_gru          models/tiny_gru.py:25-35        Window(4,3) . birnn(Gru 12->4) . FF 8->4 . Softmax
    baseline_gru      models/baseline_gru.py:24-44    Window . birnn(Gru) . FF . birnn(Gru) . FF . Softmax (size 64)
    baseline_lstm     models/baseline_lstm.py:24-44   same with peephole Lstm cells
    baseline_raw_gru  models/baseline_raw_gru.py:21-37   Conv(1->64,w11,s2,tanh) . birnn . FF . birnn . FF . Softmax
    bigger_raw_gru    models/bigger_raw_gru.py:21-37     Conv(1->32) . birnn(Gru ->96) . FF ->128 . birnn . FF . Softmax
    raw_0.98_rgrgr    models/raw_0.98_rgrgr.py:17-35     Conv(1->96,w11,s5,elu) . 5 x Gru 96 alternating direction
    raw_1.00_rGr      models/raw_1.00_rGr.py:6-23        Conv(1->128,s2,tanh) . Rev Gru 110 . Gru 142 . Rev Gru 110
    pretrained        models/pretrained.pkl              Conv(1->128,s5,elu) . Rev Gru 112 . Gru 144 . Rev Gru 112

Weights: `truncated_normal(sd)` Xavier-style init as the reference's constructors apply it (seeded through numpy's
global RandomState for reproducibility), or the trained values of `pretrained` from an .npz export.
"""
import json

import numpy as np

from . import module_tools as smt

MODEL_DEFAULTS = {
    "tiny_gru": dict(nfeature=4, winlen=3, stride=1, size=4),
    "baseline_gru": dict(nfeature=4, winlen=3, stride=1, size=64),
    "baseline_lstm": dict(nfeature=4, winlen=3, stride=1, size=64),
    "baseline_raw_gru": dict(nfeature=1, winlen=11, stride=2, size=64),
    "bigger_raw_gru": dict(nfeature=1, winlen=11, stride=2, size=(32, 96, 128)),
    "raw_0.98_rgrgr": dict(nfeature=1, winlen=11, stride=5),
    "raw_1.00_rGr": dict(nfeature=1, winlen=11, stride=2),
    "pretrained": dict(nfeature=1, winlen=11, stride=5),
}


def _bi(cell, insize, size, init, **kw):
    return smt.birnn(cell(insize, size, init=init, has_bias=True, fun=smt.tanh, **kw),
                     cell(insize, size, init=init, has_bias=True, fun=smt.tanh, **kw))


def _uni_stack(nfeature, winlen, stride, conv_size, conv_fun, gru_sizes, nstate, init):
    """conv front end + GRU layers of alternating direction, first one reversed."""
    seq = [smt.Convolution(nfeature, conv_size, winlen, stride, init=init, has_bias=True, fun=conv_fun)]
    prev = conv_size
    for i, n in enumerate(gru_sizes):
        g = smt.Gru(prev, n, init=init, has_bias=True, fun=smt.tanh)
        seq.append(smt.Reverse(g) if i % 2 == 0 else g)
        prev = n
    seq.append(smt.Softmax(prev, nstate, init=init, has_bias=True))
    return smt.Serial(seq)


def build_model(name, klen=5, sd=0.5, nbase=smt.DEFAULT_NBASE, seed=None, **overrides):
    """Network `name` with the reference factory's defaults (overridable), random-initialised."""
    if name not in MODEL_DEFAULTS:
        raise KeyError("unknown model %r (have %s)" % (name, sorted(MODEL_DEFAULTS)))
    cfg = dict(MODEL_DEFAULTS[name])
    cfg.update(overrides)
    if seed is not None:
        np.random.seed(seed)
    init = smt.partial(smt.truncated_normal, sd=sd)
    nstate = smt.nstate(klen, nbase=nbase)
    nf, w, s = cfg["nfeature"], cfg["winlen"], cfg["stride"]
    if name == "raw_0.98_rgrgr":
        return _uni_stack(nf, w, s, 96, smt.elu, [96] * 5, nstate, init)
    if name == "raw_1.00_rGr":
        return _uni_stack(nf, w, s, 128, smt.tanh, [110, 142, 110], nstate, init)
    if name == "pretrained":    # architecture of the shipped pickle, random weights (trained ones: from_weights_npz)
        return _uni_stack(nf, w, s, 128, smt.elu, [112, 144, 112], nstate, init)
    size = cfg["size"]
    if name == "tiny_gru":
        assert s == 1, "Model only supports stride of 1"
        return smt.Serial([smt.Window(nf, w), _bi(smt.Gru, nf * w, size, init),
                           smt.FeedForward(2 * size, size, has_bias=True, fun=smt.tanh),        # zero init, :31
                           smt.Softmax(size, nstate, init=init, has_bias=True)])
    if name in ("baseline_gru", "baseline_lstm"):
        assert s == 1, "Model only supports stride of 1"
        cell, kw = (smt.Gru, {}) if name == "baseline_gru" else (smt.Lstm, {"has_peep": True})
        return smt.Serial([smt.Window(nf, w), _bi(cell, nf * w, size, init, **kw),
                           smt.FeedForward(2 * size, size, has_bias=True, fun=smt.tanh),        # zero init
                           _bi(cell, size, size, init, **kw),
                           smt.FeedForward(2 * size, size, init=init, has_bias=True, fun=smt.tanh),
                           smt.Softmax(size, nstate, init=init, has_bias=True)])
    if name == "baseline_raw_gru":
        c, r, f = size, size, size
    else:                       # bigger_raw_gru
        c, r, f = size
    return smt.Serial([smt.Convolution(nf, c, w, s, init=init, has_bias=True, fun=smt.tanh),
                       _bi(smt.Gru, c, r, init),
                       smt.FeedForward(2 * r, f, has_bias=True, fun=smt.tanh),                  # zero init
                       _bi(smt.Gru, f, r, init),
                       smt.FeedForward(2 * r, f, init=init, has_bias=True, fun=smt.tanh),
                       smt.Softmax(f, nstate, init=init, has_bias=True)])


def randomise_zero_layers(net, sd=0.5, seed=1):
    """The factories leave some FeedForward layers zero-initialised (e.g. models/baseline_raw_gru.py:33), which
    makes everything downstream constant.  For benchmarking/parity on random weights give them the same
    truncated-normal init as their siblings."""
    rs = np.random.RandomState(seed)
    from scipy.stats import truncnorm

    def visit(layer):
        for sub in getattr(layer, "layers", []):
            visit(sub)
        if hasattr(layer, "layer"):
            visit(layer.layer)
        if isinstance(layer, smt.FeedForward) and not layer.W.get_value().any():
            shape = layer.W.shape
            w = sd * truncnorm.rvs(-2, 2, size=shape, random_state=rs) / np.sqrt(sum(shape))
            layer.W.set_value(w.astype(np.float32))
            layer.b.set_value((sd * truncnorm.rvs(-2, 2, size=shape[0], random_state=rs)).astype(np.float32))
    visit(net)
    return net


def from_weights_npz(path):
    """Rebuild the trained rGr network of models/pretrained.pkl from the .npz export made by
    tests/golden/make_goldens.py (arrays l<i>_<attr> + a JSON layer description)."""
    data = np.load(path)
    desc = json.loads(str(data["description_json"]))
    seq = []
    for d in desc:
        i = d["index"]
        fun = getattr(smt, d["fun"]) if "fun" in d else None

        def W(attr):
            return data["l%d_%s" % (i, attr)]
        if d["type"] == "Convolution":
            layer = smt.Convolution(d["_insize"], d["_size"], d["winlen"], d["stride"], has_bias=d["has_bias"],
                                    fun=fun, padding_mode=tuple(d["padding"]))
            layer.padding_mode = d["padding_mode"]
            layer.set_params({"W": W("W"), "b": W("b")})
        elif d["type"] == "Gru":
            n, ins = d["_size"], d["_insize"]
            layer = smt.Gru(ins, n, has_bias=d["has_bias"], fun=fun, gatefun=getattr(smt, d["gatefun"]))
            layer.set_params({"iW": W("iW").reshape(3, n, ins), "sW": W("sW").reshape(2, n, n), "sW2": W("sW2"),
                              "b": W("b").reshape(3, n)})
        elif d["type"] == "Softmax":
            layer = smt.Softmax(d["_insize"], d["_size"], has_bias=d["has_bias"])
            layer.set_params({"W": W("W"), "b": W("b")})
        else:
            raise ValueError("unexpected layer type %r in weight export" % d["type"])
        seq.append(smt.Reverse(layer) if d["reverse"] else layer)
    return smt.Serial(seq)
