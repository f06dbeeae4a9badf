"""Case definitions shared by tests/golden/make_layer_goldens.py (which runs the REFERENCE's layer code on them, in the
build container) and by the tests (which run the oracles and the HIP path on the same tensors).

A case is a JSON-able layer tree in the vocabulary of the reference's `Layer.json()` type strings
(layers.py: "serial", "parallel", "reverse", "convolution", "window", "feed-forward", "softmax_old"->"softmax", "GRU",
"LSTM").  Tensors are not stored: every weight and input is a RECIPE `{"seed", "shape", "scale"}` that both sides expand
with numpy's legacy `RandomState` (whose streams are frozen by numpy's compatibility policy); the generator records the
sha256 of what it expanded so that a drifting generator is detected instead of silently comparing different inputs.
Weights are laid out exactly as the reference's `step`/`run` code reads them (they are written straight into the shared
variables, the way a model pickle restores them -- never through `set_params`, whose Lstm layout differs).
"""
import hashlib

import numpy as np


def recipe(seed, shape, scale):
    return {"seed": int(seed), "shape": [int(s) for s in shape], "scale": float(scale)}


def expand(r, dtype=np.float32):
    """Recipe -> array: uniform(-1, 1) * scale, rounded to float32 (then optionally widened)."""
    a = np.random.RandomState(r["seed"]).uniform(-1.0, 1.0, size=r["shape"]) * r["scale"]
    return a.astype(np.float32).astype(dtype)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


_PARAM_KEYS = {"GRU": ("iW", "sW", "sW2", "b"), "LSTM": ("iW", "sW", "b", "p"), "convolution": ("W", "b"),
               "feed-forward": ("W", "b"), "softmax": ("W", "b")}


def param_keys(node):
    return _PARAM_KEYS.get(node["type"], ())


def walk(node):
    """Leaf layers in `params()` order (Serial / Parallel concatenate, Reverse passes through: layers.py:1440,1476,1546)."""
    t = node["type"]
    if t in ("serial", "parallel"):
        for sub in node["sublayers"]:
            yield from walk(sub)
    elif t == "reverse":
        yield from walk(node["sublayer"])
    else:
        yield node


def materialise(node, dtype=np.float32):
    """Recipe tree -> the `spec` dicts oracle.run_network / oracle_np.run_network take (arrays instead of recipes)."""
    t = node["type"]
    if t in ("serial", "parallel"):
        return {"type": t, "sublayers": [materialise(s, dtype) for s in node["sublayers"]]}
    if t == "reverse":
        return {"type": t, "sublayer": materialise(node["sublayer"], dtype)}
    out = dict(node)
    for k in param_keys(node):
        out[k] = None if node.get(k) is None else expand(node[k], dtype)
    if t == "convolution":
        out["padding"] = tuple(node["padding"])
    return out


def param_arrays(node, dtype=np.float32):
    """Arrays in `network.params()` order (parameters a layer does not have -- has_bias False -- are skipped)."""
    return [expand(leaf[k], dtype) for leaf in walk(node) for k in param_keys(leaf) if leaf.get(k) is not None]


# --------------------------------------------------------------------------------------- layer-level cases
def _seeds(base):
    n = [base * 1000]

    def nxt():
        n[0] += 1
        return n[0]
    return nxt


def gru(seed, I, n, bias=True, act="tanh", gate="sigmoid", wscale=1.0):
    s = _seeds(seed)
    return {"type": "GRU", "insize": I, "size": n, "activation": act, "gate": gate,
            "iW": recipe(s(), (3 * n, I), 1.5 * wscale / np.sqrt(I)), "sW": recipe(s(), (2 * n, n), 2.0 * wscale / np.sqrt(n)),
            "sW2": recipe(s(), (n, n), 2.0 * wscale / np.sqrt(n)), "b": recipe(s(), (3 * n,), 0.5) if bias else None}


def lstm(seed, I, n, bias=True, peep=True, act="tanh", gate="sigmoid"):
    s = _seeds(seed)
    return {"type": "LSTM", "insize": I, "size": n, "activation": act, "gate": gate,
            "iW": recipe(s(), (4 * n, I), 1.5 / np.sqrt(I)), "sW": recipe(s(), (4 * n, n), 2.0 / np.sqrt(n)),
            "b": recipe(s(), (4 * n,), 0.5) if bias else None, "p": recipe(s(), (3, n), 0.7) if peep else None}


def conv(seed, Cin, Cout, w, stride, mode="same", act="tanh", bias=True):
    s = _seeds(seed)
    return {"type": "convolution", "insize": Cin, "size": Cout, "winlen": w, "stride": stride, "padding_mode": mode,
            "activation": act, "W": recipe(s(), (Cout, Cin, w), 1.2 / np.sqrt(Cin * w)),
            "b": recipe(s(), (Cout,), 0.5) if bias else None}


def ff(seed, I, n, act="tanh", bias=True):
    s = _seeds(seed)
    return {"type": "feed-forward", "insize": I, "size": n, "activation": act,
            "W": recipe(s(), (n, I), 1.5 / np.sqrt(I)), "b": recipe(s(), (n,), 0.5) if bias else None}


def softmax(seed, I, n, bias=True):
    s = _seeds(seed)
    return {"type": "softmax", "insize": I, "size": n, "W": recipe(s(), (n, I), 3.0 / np.sqrt(I)),
            "b": recipe(s(), (n,), 0.5) if bias else None}


def rev(node):
    return {"type": "reverse", "sublayer": node}


def par(*nodes):
    return {"type": "parallel", "sublayers": list(nodes)}


def ser(*nodes):
    return {"type": "serial", "sublayers": list(nodes)}


def window(I, w):
    return {"type": "window", "insize": I, "w": w}


def layer_cases():
    """name -> (tree, input recipe).  Sizes cover every fused-kernel instantiation of csrc/gru_fused.hip (96->96, 64->64,
    32->96, 128->96, 64->96, 16->16, 48->32), the two-kernel sizes of models/pretrained.pkl (112, 144), the zero-padded
    twins of raw_1.00_rGr (110, 142), the MFMA Lstm sizes and the generic kernels."""
    c = {}

    def add(name, tree, T, B, xscale=1.0):
        insize = next(walk(tree))["insize"] if tree["type"] != "window" else tree["insize"]
        c[name] = {"tree": tree, "x": recipe(900000 + len(c), (T, B, insize), xscale)}

    for i, (I, n) in enumerate([(5, 8), (16, 16), (48, 32), (64, 64), (96, 96), (32, 96), (128, 96), (64, 96), (128, 112),
                                (112, 144), (128, 110), (110, 142)]):
        T, B = (12, 3) if n <= 16 else (10, 5)
        add("gru_%d_%d" % (I, n), gru(10 + i, I, n), T, B)
        add("gru_%d_%d_rev" % (I, n), rev(gru(40 + i, I, n)), T, B)
    add("gru_nobias_7_12", gru(70, 7, 12, bias=False), 9, 2)
    add("gru_strong_96_96", gru(71, 96, 96, wscale=3.0), 14, 4, xscale=2.0)       # saturating gates, |w| like trained models
    add("gru_relu_sigmoidpm_6_8", gru(72, 6, 8, act="relu", gate="sigmoid_pm", wscale=0.5), 8, 2)
    for i, (I, n, bias, peep) in enumerate([(7, 16, True, True), (12, 64, True, True), (64, 64, True, False),
                                            (5, 12, False, False), (12, 96, True, True)]):
        add("lstm_%d_%d_%d%d" % (I, n, bias, peep), lstm(100 + i, I, n, bias, peep), 9, 3)
        add("lstm_%d_%d_%d%d_rev" % (I, n, bias, peep), rev(lstm(120 + i, I, n, bias, peep)), 9, 3)
    # Convolution: every padding mode of conv.calculate_padding, the strides / widths / activations models/*.py use
    for i, (Cin, Cout, w, s, mode, act, T) in enumerate([
            (1, 96, 11, 5, "same", "elu", 83), (1, 64, 11, 2, "same", "tanh", 41), (1, 32, 11, 2, "same", "tanh", 40),
            (1, 128, 11, 5, "same", "elu", 100), (1, 128, 11, 2, "same", "tanh", 37), (1, 8, 11, 1, "same", "tanh", 30),
            (12, 32, 11, 5, "same", "tanh", 100), (3, 4, 4, 1, "same", "linear", 30), (2, 3, 5, 3, "valid", "relu", 31),
            (2, 5, 5, 2, "full", "tanh", 20), (3, 6, 6, 2, "half", "tanh", 25), (2, 4, 4, 1, "same_left", "tanh", 17),
            (2, 4, 7, 3, 2, "elu", 29), (1, 6, 9, 4, 0, "tanh", 33)]):       # tuple modes: unreachable in the reference on py3 (conv.py:47-49)
        add("conv_%d" % i, conv(200 + i, Cin, Cout, w, s, mode, act), T, 3 if Cout < 64 else 2, xscale=1.5)
    add("conv_nobias", conv(230, 1, 8, 11, 5, "same", "elu", bias=False), 50, 2)
    add("window_3", window(4, 3), 11, 3)
    add("window_5", window(2, 5), 9, 2)
    add("ff_tanh", ff(240, 12, 7), 6, 3)
    add("ff_linear_nobias", ff(241, 9, 5, act="linear", bias=False), 6, 3)
    add("softmax_65", softmax(242, 10, 65), 7, 3)
    add("softmax_1025", softmax(243, 96, 1025), 3, 2)
    add("birnn_gru", par(gru(250, 6, 8), rev(gru(251, 6, 8))), 11, 3)
    add("birnn_lstm", par(lstm(252, 6, 8), rev(lstm(253, 6, 8))), 11, 3)
    add("serial_window_birnn_ff_softmax", ser(window(4, 3), par(gru(260, 12, 8), rev(gru(261, 12, 8))), ff(262, 16, 8),
                                              softmax(263, 8, 65)), 13, 2)
    add("serial_conv_rgr_softmax", ser(conv(270, 1, 16, 11, 5, "same", "elu"), rev(gru(271, 16, 16)), gru(272, 16, 16),
                                       rev(gru(273, 16, 16)), softmax(274, 16, 65)), 60, 3, xscale=1.7)
    return c


# name -> (factory file under /root/reference/models, kwargs the factory is called with, input [T, B, F] and scale).
# klen 5 (1025 states) as bin/basecall_network.py:32 defaults; everything else is each factory's own default.
MODEL_CASES = {
    "tiny_gru": ("tiny_gru.py", dict(klen=5, sd=0.5, nfeature=4, winlen=3, stride=1), (14, 2, 4), 1.0),
    "baseline_gru": ("baseline_gru.py", dict(klen=5, sd=0.5, nfeature=4, winlen=3, stride=1), (12, 2, 4), 1.0),
    "baseline_lstm": ("baseline_lstm.py", dict(klen=5, sd=0.5, nfeature=4, winlen=3, stride=1), (12, 2, 4), 1.0),
    "baseline_raw_gru": ("baseline_raw_gru.py", dict(klen=5, sd=0.5, nfeature=1, winlen=11, stride=2), (24, 2, 1), 1.7),
    "bigger_raw_gru": ("bigger_raw_gru.py", dict(klen=5, sd=0.5, nfeature=1, winlen=11, stride=2), (24, 2, 1), 1.7),
    "raw_0.98_rgrgr": ("raw_0.98_rgrgr.py", dict(klen=5, sd=0.5, nfeature=1, winlen=11, stride=5), (63, 2, 1), 1.7),
    "raw_1.00_rGr": ("raw_1.00_rGr.py", dict(klen=5, sd=0.5, nfeature=1, winlen=11, stride=2), (24, 2, 1), 1.7),
}
