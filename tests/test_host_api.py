"""Host-side logic of the API mirror (no GPU needed): layer bookkeeping, model files, k-mer strings, sharding."""
import json
import os
import pickle

import numpy as np
import pytest

from tests.conftest import GOLDEN

REF = "/root/reference"


def test_layer_json_and_params_roundtrip():
    from sloika_amd import layers, activation
    rs = np.random.RandomState(0)
    g = layers.Gru(5, 4, has_bias=True)
    vals = {"iW": rs.normal(size=(3, 4, 5)).astype(np.float32), "sW": rs.normal(size=(2, 4, 4)).astype(np.float32),
            "sW2": rs.normal(size=(4, 4)).astype(np.float32), "b": rs.normal(size=(3, 4)).astype(np.float32)}
    g.set_params(vals)
    j = g.json(params=True)
    assert list(j.keys()) == ['type', 'activation', 'gate', 'size', 'insize', 'bias', 'params']      # layers.py:985-996
    assert j['type'] == "GRU" and j['activation'] == 'tanh' and j['gate'] == 'sigmoid'
    for k in vals:
        assert np.array_equal(np.asarray(j['params'][k], dtype=np.float32), vals[k])
    assert [p.shape for p in g.params()] == [(12, 5), (8, 4), (4, 4), (12,)]                      # layers.py:979-983
    with pytest.raises(AssertionError):
        g.set_params(dict(vals, sW2=np.zeros((3, 4), dtype=np.float32)))                           # layers.py:1007
    l = layers.Lstm(5, 4, has_bias=True, has_peep=True)
    b = rs.normal(size=(4, 4)).astype(np.float32)
    l.set_params({"iW": np.zeros((4, 4, 5), np.float32), "sW": np.zeros((4, 4, 4), np.float32), "b": b,
                  "p": np.zeros((3, 4), np.float32)})
    assert np.array_equal(l.b.get_value(), b.transpose().reshape(-1))                              # layers.py:666
    assert np.array_equal(layers.Lstm(3, 2, has_bias=True).b.get_value(), [0, 0, 0, 0, 2, 2, 0, 0]) # layers.py:637
    c = layers.Convolution(1, 8, 11, 5, fun=activation.elu)
    assert c.padding == (5, 5) and c.json()['activation'] == 'elu' and c.json()['padding_mode'] == 'same'
    assert layers.Window(4, 3).size == 12
    with pytest.raises(AssertionError):
        layers.Serial([layers.FeedForward(3, 4), layers.FeedForward(5, 2)])                        # layers.py:1537-1538
    with pytest.raises(AssertionError):
        layers.Parallel([])
    bi = layers.birnn(layers.Gru(3, 4), layers.Gru(3, 4))
    assert bi.size == 8 and bi.json()['sublayers'][1]['type'] == 'reverse'
    with pytest.raises(NotImplementedError):
        layers.Scrn(3, 4)


def test_calculate_padding():
    from sloika_amd import conv
    assert conv.calculate_padding('same', 11) == (5, 5) and conv.calculate_padding('same', 4) == (1, 2)
    assert conv.calculate_padding('same_left', 4) == (2, 1) and conv.calculate_padding('half', 4) == (2, 2)
    assert conv.calculate_padding('valid', 7) == (0, 0) and conv.calculate_padding('full', 7) == (6, 6)
    assert conv.calculate_padding(3, 7) == (3, 3) and conv.calculate_padding((1, 2), 7) == (1, 2)
    with pytest.raises(AssertionError):
        conv.calculate_padding('bogus', 3)


def test_variables():
    from sloika_amd import variables as sv
    assert sv.nkmer(5) == 1024 and sv.nstate(5) == 1025 and sv.nstate(3, nbase=5) == 126
    assert sv.nstate(5, transducer=False, bad_state=False) == 1024


def test_model_table_matches_counts():
    """Parameter counts of SURVEY.md 8(d): 182 977 / 385 921 / 378 497."""
    from sloika_amd import models
    counts = {n: sum(p.get_value().size for p in models.build_model(n, seed=1).params())
              for n in ("baseline_raw_gru", "bigger_raw_gru", "raw_0.98_rgrgr")}
    assert counts == {"baseline_raw_gru": 182977, "bigger_raw_gru": 385921, "raw_0.98_rgrgr": 378497}
    net = models.from_weights_npz(os.path.join(GOLDEN, "pretrained_weights.npz"))
    assert [(type(l).__name__, l.insize, l.size) for l in net.layers] == [
        ('Convolution', 1, 128), ('Reverse', 128, 112), ('Gru', 112, 144), ('Reverse', 144, 112), ('Softmax', 112, 1025)]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_reference_model_files_load_unchanged():
    """models/*.py run unchanged through the `sloika` alias; models/pretrained.pkl loads without Theano."""
    from sloika_amd import helpers, models
    for name in models.MODEL_DEFAULTS:
        if name == "pretrained":            # a pickle, not a factory: checked below
            continue
        np.random.seed(3)
        ref = helpers.load_factory(os.path.join(REF, "models", name + ".py"), klen=5, sd=0.5)
        mine = models.build_model(name, seed=3)
        assert json.dumps(ref.json(params=True)) == json.dumps(mine.json(params=True)), name
    net = helpers.load_pickle(os.path.join(REF, "models", "pretrained.pkl"))
    exp = models.from_weights_npz(os.path.join(GOLDEN, "pretrained_weights.npz"))
    assert json.dumps(net.json(params=True), default=int) == json.dumps(exp.json(params=True), default=int)


def test_model_pickle_roundtrip(tmp_path):
    from sloika_amd import helpers, models
    net = models.build_model("raw_0.98_rgrgr", seed=5)
    path = tmp_path / "model_final.pkl"
    with open(path, "wb") as fh:
        pickle.dump(net, fh, protocol=pickle.HIGHEST_PROTOCOL)          # bin/train_network.py:145-152
    back = helpers.load_model(str(path))
    assert json.dumps(back.json(params=True)) == json.dumps(net.json(params=True))
    assert back.layers[1].layer.fun.__name__ == "tanh"


def test_bio_goldens(golden_bio, golden_decode):
    from sloika_amd import bio
    assert bio.all_kmers(3)[:8] == golden_bio["all_kmers_3_head"]
    assert bio.all_kmers(2, b"AC") == [b"AA", b"AC", b"CA", b"CC"]
    kmers = bio.all_kmers(5)
    for case in golden_bio["cases"]:
        kp = [kmers[i] for i in golden_decode["path_" + case["name"]]]
        assert bio.kmers_to_sequence(kp, always_move=True) == case["seq_always_move"]
        assert bio.kmers_to_sequence(kp, always_move=False) == case["seq_allow_stay"]
    k = golden_bio["kat"]
    assert bio.kmers_to_sequence(k["kmers"], always_move=True) == k["always_move"]
    assert bio.kmers_to_sequence(k["kmers"], always_move=False) == k["allow_stay"]
    assert bio.seq_to_kmers('ATATGCG', 3) == ['ATA', 'TAT', 'ATG', 'TGC', 'GCG']              # bio.py:149
    assert bio.max_overlap(['AAC', 'ACT', 'ACT']) == [1, 0]


def test_util(golden_transducer):
    from sloika_amd import util
    assert np.array_equal(util.geometric_prior(30, 2.0), golden_transducer["geometric_prior_30_2"])
    assert np.array_equal(util.geometric_prior(30, 2.0, rev=True), golden_transducer["geometric_prior_30_2_rev"])
    x = np.arange(20)
    assert np.array_equal(util.trim_array(x, 2, 3), x[2:-3]) and np.array_equal(util.trim_array(x, 0, 0), x)


def test_synthetic_chunks_deterministic():
    from sloika_amd import pipeline
    a = pipeline.synthetic_chunks(3, chunk_len=500, seed=7)
    b = pipeline.synthetic_chunks(2, chunk_len=500, seed=7, first_chunk=1)
    assert a.dtype == np.float32 and a.shape == (3, 500) and np.array_equal(a[1:], b)


def test_shard_bounds_cover_everything():
    from sloika_amd import shard
    for n in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_bounds(4, 2, 2)


def test_states_to_sequence_equals_string_route(golden_bio, golden_decode):
    """The integer route (state numbers, no strings) gives what the reference's string route gave on the decode goldens."""
    from sloika_amd import bio
    for case in golden_bio["cases"]:
        path = golden_decode["path_" + case["name"]]
        assert bio.states_to_sequence(path, 5, "ACGT", always_move=True) == case["seq_always_move"]
        assert bio.states_to_sequence(path, 5, b"ACGT", always_move=False) == case["seq_allow_stay"]
    assert bio.states_to_sequence([], 5) == ""
    assert bio.states_to_sequence([7], 3, "ACGT") == "ACT"
    # reference known answers, test/unit/test_bio.py:137-148
    assert bio.kmers_to_sequence(["AAC", "ACT", "CTG"]) == "AACTG"
    assert bio.max_overlap(["AAC", "ACT", "ACT", "CTG", "GGA"]) == [1, 0, 1, 2]
    assert bio.max_overlap(["AAC", "TTT"]) == [3]
    assert bio.max_overlap(["AAA", "AAA"], allow_identical=False) == [1]
    assert bio.moves_compatible(["AAC", "ACT", "GGA"], [1, 3]) == [True, True]
    assert bio.moves_compatible(["AAC", "ACT"], [2]) == [False]


def test_model_unpickler_refuses_code_execution(tmp_path):
    """A model file is data: globals outside theano.* / sloika.* / numpy's array rebuilders are refused."""
    import pickle
    from sloika_amd import helpers

    class Evil(object):
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))
    p = tmp_path / "evil.pkl"
    with open(p, "wb") as fh:
        pickle.dump(Evil(), fh)
    with pytest.raises(pickle.UnpicklingError):
        helpers.load_pickle(str(p))
    assert not (tmp_path / "pwned").exists()


def test_model_unpickler_refuses_dotted_and_foreign_package_globals(tmp_path):
    """Protocol >= 4 resolves a dotted global attribute by attribute: ('sloika_amd.build', 'subprocess.getoutput') would reach
    any module this package imports.  Dotted names, modules other than layers / activation, and objects of those modules that
    are not layer classes or activation functions are all refused; the legitimate globals still load."""
    import io
    import pickle
    import pickletools  # noqa: F401
    from sloika_amd import helpers, layers

    def crafted(module, name, arg):
        # PROTO 4, SHORT_BINUNICODE module, SHORT_BINUNICODE name, STACK_GLOBAL, SHORT_BINUNICODE arg, TUPLE1, REDUCE, STOP
        def su(x):
            b = x.encode()
            return b"\x8c" + bytes([len(b)]) + b
        return b"\x80\x04" + su(module) + su(name) + b"\x93" + su(arg) + b"\x85R."

    marker = tmp_path / "pwned"
    for module, name in (("sloika_amd.build", "subprocess.getoutput"), ("sloika_amd._lib", "os.system"),
                         ("sloika.layers", "os.system"), ("sloika_amd.layers", "np.load"), ("sloika_amd.build", "build"),
                         ("sloika_amd.layers", "reduce"), ("sloika_amd.activation", "np"), ("sloika_amd.helpers", "load_pickle")):
        with pytest.raises(pickle.UnpicklingError):
            helpers._Unpickler(io.BytesIO(crafted(module, name, "touch %s" % marker))).load()
    assert not marker.exists()
    with pytest.raises(ValueError):
        helpers._Unpickler(io.BytesIO(crafted("sloika_amd.activation", "_lookup", "np"))).load()
    # what model files do name
    assert helpers._Unpickler(io.BytesIO(crafted("sloika_amd.activation", "_lookup", "tanh"))).load().__name__ == "tanh"
    net = layers.Serial([layers.FeedForward(4, 3, has_bias=True, fun=__import__("sloika_amd").activation.tanh), layers.Softmax(3, 5)])
    clone = helpers._Unpickler(io.BytesIO(pickle.dumps(net, protocol=4))).load()
    assert isinstance(clone, layers.Serial) and clone.layers[0].fun.__name__ == "tanh"
