"""Pin the oracle's slip_update / map_to_sequence against reference-generated goldens.

Reference: sloika/viterbi_helpers.pyx:12-35, sloika/transducer.py:14-73, test/unit/test_viterbi.py:14-33.
"""
import numpy as np

from tests.conftest import regen_post


def test_slip_update_goldens(oracle, golden_transducer):
    g = golden_transducer
    for n in (3, 4, 10, 400):
        for slip in (0.0, 5.0):
            fs, fp = oracle.slip_update(g["slip_x_%d" % n], slip)
            assert np.array_equal(fs, g["slip_fs_%d_%g" % (n, slip)])
            assert np.array_equal(fp, g["slip_fp_%d_%g" % (n, slip)])
            assert fs.dtype == np.float32 and fp.dtype == np.int64
    fs, fp = oracle.slip_update(g["slip_x_tie"], 0.0)
    assert np.array_equal(fs, g["slip_fs_tie"]) and np.array_equal(fp, g["slip_fp_tie"])


def test_slip_update_same_as_python_loop(oracle):
    # restates test/unit/test_viterbi.py:14-33 (seed 0xdeadbeef, n=10, slip 5.0)
    np.random.seed(0xdeadbeef)
    x = np.random.normal(size=10).astype(np.float32)
    slip = 5.0
    y1s, y1i = oracle.slip_update(x, slip)
    y2s = np.zeros(len(x), dtype=np.float32)
    y2i = np.zeros(len(x), dtype=np.int64)
    y2s[0] = y2s[1] = -1e38
    y2s[2] = x[0] - slip
    for j in range(3, len(x)):
        if y2s[j - 1] >= x[j - 2]:
            y2s[j], y2i[j] = y2s[j - 1], y2i[j - 1]
        else:
            y2s[j], y2i[j] = x[j - 2], j - 2
        y2s[j] -= slip
    np.testing.assert_almost_equal(y1s, y2s)
    np.testing.assert_equal(y1i, y2i)


def _map_input(case, g):
    key = "map_trans_" + case["name"]
    if key in g:
        return g[key]
    gen = case["gen"]
    if "key" in gen:
        return g[gen["key"]]
    assert gen["kind"] == "prepare_post(dirichlet)"
    post = np.random.RandomState(gen["seed"]).dirichlet(np.ones(gen["nst"]) * gen["alpha"],
                                                        size=gen["nev"]).astype(np.float32)
    return np.float32(1e-5) + np.float32(1.0 - 1e-5) * post


def test_map_to_sequence_goldens(oracle, golden_cases, golden_transducer):
    import hashlib
    g = golden_transducer
    for case in golden_cases["map_cases"]:
        trans = _map_input(case, g)
        assert hashlib.sha256(np.ascontiguousarray(trans).tobytes()).hexdigest() == case["sha256"], case["name"]
        name = case["name"]
        pi = g["map_pi_" + name] if case["has_pi"] else None
        pf = g["map_pf_" + name] if case["has_pf"] else None
        score, path = oracle.map_to_sequence(trans, g["map_seq_" + name], slip=case["slip"], prior_initial=pi,
                                             prior_final=pf, log=case["log"])
        assert np.array_equal(path, g["map_path_" + name]), name
        assert float(score) == float.fromhex(case["score_hex"]), name
