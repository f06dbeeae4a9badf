"""GPU parity: slip_update and map_to_sequence through the C ABI, bit exact against reference goldens."""
import hashlib
import zlib

import numpy as np
import pytest

from tests.gpu_util import need_gpu
from tests.test_oracle_transducer import _map_input

pytestmark = pytest.mark.gpu


def test_slip_update_goldens(golden_transducer):
    need_gpu()
    from sloika_amd import viterbi_helpers
    g = golden_transducer
    for n in (3, 4, 10, 400):
        for slip in (0.0, 5.0):
            fs, fp = viterbi_helpers.slip_update(g["slip_x_%d" % n], slip)
            assert fs.dtype == np.float32 and fp.dtype == np.int64
            assert np.array_equal(fs, g["slip_fs_%d_%g" % (n, slip)]) and np.array_equal(fp, g["slip_fp_%d_%g" % (n, slip)])
    fs, fp = viterbi_helpers.slip_update(g["slip_x_tie"], 0.0)
    assert np.array_equal(fs, g["slip_fs_tie"]) and np.array_equal(fp, g["slip_fp_tie"])
    with pytest.raises(ValueError):
        viterbi_helpers.slip_update(np.zeros(2, dtype=np.float32), 1.0)


def test_map_to_sequence_goldens(golden_cases, golden_transducer):
    need_gpu()
    from sloika_amd import transducer
    g = golden_transducer
    for case in golden_cases["map_cases"]:
        trans = _map_input(case, g)
        name = case["name"]
        if not case["log"]:
            trans = np.log(trans)          # feed the same float32 logs the reference computed (transducer.py:30)
        pi = g["map_pi_" + name] if case["has_pi"] else None
        pf = g["map_pf_" + name] if case["has_pf"] else None
        score, path = transducer.map_to_sequence(trans, g["map_seq_" + name], slip=case["slip"], prior_initial=pi,
                                                 prior_final=pf, log=True)
        assert np.array_equal(path, g["map_path_" + name]), name
        assert float(score) == float.fromhex(case["score_hex"]), name


def test_map_to_sequence_device_log_close(golden_cases, golden_transducer):
    need_gpu()
    from sloika_amd import transducer
    g = golden_transducer
    case = [c for c in golden_cases["map_cases"] if c["name"] == "m100_post"][0]
    score, path = transducer.map_to_sequence(g["map_trans_m100_post"], g["map_seq_m100_post"], slip=5.0, log=False)
    assert float(score) == pytest.approx(float.fromhex(case["score_hex"]), rel=1e-5)
    assert (path == g["map_path_m100_post"]).mean() > 0.95
    with pytest.raises(ValueError):
        transducer.map_to_sequence(g["map_trans_m100_post"], g["map_seq_m100_post"], slip=None)


def test_map_to_sequence_batch_equals_goldens_and_single_calls(oracle, golden_cases, golden_transducer):
    """Ragged batch in one launch: every read must reproduce the reference's golden path and score bit for bit
    (cases without priors share a batch; a second batch carries priors for every read)."""
    need_gpu()
    from sloika_amd import transducer
    g = golden_transducer
    plain = [c for c in golden_cases["map_cases"] if not c["has_pi"] and not c["has_pf"]]
    by_slip = {}
    for c in plain:
        by_slip.setdefault((c["slip"], _map_input(c, g).shape[1]), []).append(c)
    checked = 0
    for (slip, nst), cases in by_slip.items():
        trans = []
        for c in cases:
            t = _map_input(c, g)
            trans.append(t if c["log"] else np.log(t))
        seqs = [g["map_seq_" + c["name"]] for c in cases]
        scores, paths = transducer.map_to_sequence_batch(trans, seqs, slip)
        for c, sc, pa in zip(cases, scores, paths):
            assert np.array_equal(pa, g["map_path_" + c["name"]]), c["name"]
            assert float(sc) == float.fromhex(c["score_hex"]), c["name"]
            checked += 1
    assert checked == len(plain) and checked >= 2
    # ragged random batch with priors vs the oracle, read by read
    rs = np.random.RandomState(11)
    trans, seqs, pis, pfs = [], [], [], []
    for nev, npos in ((40, 17), (3, 3), (120, 64), (75, 9)):
        trans.append(np.log(rs.dirichlet(np.ones(65) * 0.4, size=nev)).astype(np.float32))
        seqs.append(rs.randint(1, 65, size=npos).astype(np.int32))
        pis.append(rs.normal(size=npos) * 2.0)
        pfs.append(rs.normal(size=npos) * 2.0)
    scores, paths = transducer.map_to_sequence_batch(trans, seqs, 2.5, prior_initial=pis, prior_final=pfs)
    for b in range(len(trans)):
        o_score, o_path = oracle.map_to_sequence(trans[b], seqs[b], 2.5, prior_initial=pis[b], prior_final=pfs[b])
        assert np.array_equal(paths[b], o_path) and float(scores[b]) == float(o_score)
        s1, p1 = transducer.map_to_sequence(trans[b], seqs[b], slip=2.5, prior_initial=pis[b], prior_final=pfs[b])
        assert np.array_equal(paths[b], p1) and float(scores[b]) == float(s1)
    with pytest.raises(ValueError):
        transducer.map_to_sequence_batch(trans, seqs[:2], 2.5)


def _slip_inputs(rs, n, kind):
    if kind == "noise":
        return (rs.normal(size=n) * 8 - 200).astype(np.float32)
    if kind == "falling":                      # every value below the decayed chain: one chain owns the whole array
        return (-np.arange(n) * 7.0 - rs.uniform(0, 1, size=n)).astype(np.float32)
    if kind == "rising":                       # every value beats the chain
        return (np.arange(n) * 0.37 - 900 + rs.uniform(0, 0.1, size=n)).astype(np.float32)
    if kind == "flat":
        return np.full(n, -123.456, dtype=np.float32)
    if kind == "ridge":                        # remap-like: a peak, slow decay to the right (about the slip rate), cliffs
        peak = rs.randint(0, n)
        x = -np.abs(np.arange(n) - peak) * rs.choice([1.0, 4.9, 5.0, 5.1]) - 300 + rs.normal(size=n) * 0.7
        x[rs.randint(0, n, size=max(1, n // 40))] -= 400
        return x.astype(np.float32)
    if kind == "ties":                         # small integers: decayed values collide exactly all the time
        return rs.randint(-12, 0, size=n).astype(np.float32) * 2.5
    if kind == "neginf":
        x = (rs.normal(size=n) * 3 - 50).astype(np.float32)
        x[rs.uniform(size=n) < 0.3] = -np.inf
        if rs.uniform() < 0.5:
            x[0] = -np.inf
        return x
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["noise", "falling", "rising", "flat", "ridge", "ties", "neginf"])
def test_slip_update_wave_scan_is_the_sequential_recurrence(oracle, kind):
    """The 64-lane evaluation (csrc/transducer.hip slip_scan_wave) against the sequential oracle, bit for bit, on inputs
    built to exercise every branch: chains that die at once, chains that survive many segments, exact ties, -inf."""
    need_gpu()
    from sloika_amd import viterbi_helpers
    rs = np.random.RandomState(zlib.crc32(kind.encode()) % 10000)
    for n in [3, 4, 5, 6, 63, 64, 65, 66, 67, 129, 130, 131, 194, 257, 500, 1000, 2047, 4099, 20000]:
        for slip in (5.0, 2.5, 0.1, 0.0, 37.25):
            x = _slip_inputs(rs, n, kind)
            fs, fp = viterbi_helpers.slip_update(x, slip)
            wfs, wfp = oracle.slip_update(x, slip)
            assert np.array_equal(fs.view(np.uint32), np.asarray(wfs, dtype=np.float32).view(np.uint32)), (kind, n, slip)
            assert np.array_equal(fp, wfp), (kind, n, slip)
