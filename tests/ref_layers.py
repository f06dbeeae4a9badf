"""Access to tests/golden/layers.npz + layers_cases.json: outputs of the reference's own sloika/layers.py, conv.py,
models/*.py, basecall.py and bin/train_network.py:wrap_network, produced by tests/golden/make_layer_goldens.py."""
import json
import os
import sys

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)
import layer_cases as lc  # noqa: E402

_cache = {}


def meta():
    if "meta" not in _cache:
        with open(os.path.join(GOLDEN, "layers_cases.json")) as fh:
            _cache["meta"] = json.load(fh)
    return _cache["meta"]


def arrays():
    if "npz" not in _cache:
        _cache["npz"] = np.load(os.path.join(GOLDEN, "layers.npz"))
    return _cache["npz"]


def check_inputs(case):
    """The tensors expanded here are the ones the reference saw."""
    h = [lc.sha(lc.expand(case["x"]))] + [lc.sha(a) for a in lc.param_arrays(case["tree"])]
    assert lc.sha(np.frombuffer("".join(h).encode(), dtype=np.uint8).astype(np.float32)) == case["sha256"], \
        "regenerated tensors differ from the ones the reference was run on"


def build_amd(node):
    """Recipe tree -> this package's layer objects (sloika_amd.layers), weights written as a model pickle would."""
    from sloika_amd import activation, layers
    t = node["type"]
    if t == "serial":
        return layers.Serial([build_amd(s) for s in node["sublayers"]])
    if t == "parallel":
        return layers.Parallel([build_amd(s) for s in node["sublayers"]])
    if t == "reverse":
        return layers.Reverse(build_amd(node["sublayer"]))
    if t == "window":
        return layers.Window(node["insize"], node["w"])
    bias = node.get("b") is not None
    if t == "GRU":
        layer = layers.Gru(node["insize"], node["size"], has_bias=bias, fun=getattr(activation, node["activation"]),
                           gatefun=getattr(activation, node["gate"]))
    elif t == "LSTM":
        layer = layers.Lstm(node["insize"], node["size"], has_bias=bias, has_peep=node.get("p") is not None,
                            fun=getattr(activation, node["activation"]), gatefun=getattr(activation, node["gate"]))
    elif t == "convolution":
        mode = node["padding_mode"]
        layer = layers.Convolution(node["insize"], node["size"], node["winlen"], node["stride"], has_bias=bias,
                                   fun=getattr(activation, node["activation"]), padding_mode=mode)
        assert tuple(layer.padding) == tuple(node["padding"]), "calculate_padding(%r) differs from the reference's" % (mode,)
    elif t == "feed-forward":
        layer = layers.FeedForward(node["insize"], node["size"], has_bias=bias, fun=getattr(activation, node["activation"]))
    elif t == "softmax":
        layer = layers.Softmax(node["insize"], node["size"], has_bias=bias)
    else:
        raise ValueError(t)
    for k in lc.param_keys(node):
        if node.get(k) is not None:
            getattr(layer, k).set_value(lc.expand(node[k]))
    return layer


def int_edit_distance(a, b):
    """Levenshtein distance between two integer sequences (row-wise numpy DP)."""
    a, b = np.asarray(a), np.asarray(b)
    idx = np.arange(len(b) + 1)
    prev = idx.copy()
    for i, ca in enumerate(a, 1):
        cur = np.concatenate(([i], np.minimum(prev[:-1] + (b != ca), prev[1:] + 1)))
        prev = np.minimum.accumulate(cur - idx) + idx
    return int(prev[-1])
