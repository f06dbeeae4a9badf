"""Pin the CPU oracle's decode functions against the reference's own KATs and reference-generated goldens.

Reference: sloika/decode.py:21-93, test/unit/test_decode.py:233-256.
"""
import numpy as np
import pytest

from tests.conftest import decode_case_input


def test_viterbi_reference_kats(oracle, golden_decode):
    # test/unit/test_decode.py:233-236
    score, path = oracle.viterbi(golden_decode["kat_post3"], 3)
    assert score == pytest.approx(-11.130084569094556, abs=1e-7)
    assert path == [49, 7, 63, 63]
    # test/unit/test_decode.py:238-241
    score, path = oracle.viterbi(golden_decode["kat_post3"], 3, skip_pen=3.0)
    assert score == pytest.approx(-11.936803444063674, abs=1e-7)
    assert path == [49, 7, 31, 63, 63]
    # test/unit/test_decode.py:244-256 (5-base alphabet)
    score, path = oracle.viterbi(golden_decode["kat_mod_post"], 3, skip_pen=5.0, nbase=5)
    assert path == [int(x) - 1 for x in golden_decode["kat_mod_seq"] if x]


def test_viterbi_goldens_bit_exact(oracle, golden_cases, golden_decode):
    for case in golden_cases["decode_cases"]:
        post = decode_case_input(case, golden_decode)
        score, path = oracle.viterbi(post, case["klen"], skip_pen=case["skip_pen"], log=case["log"],
                                     nbase=case["nbase"])
        assert list(golden_decode["path_" + case["name"]]) == path, case["name"]
        assert float(score) == float.fromhex(case["score_hex"]), case["name"]


def test_viterbi_python_loop_agrees_on_small_cases(oracle, golden_cases, golden_decode):
    from oracle import oracle_np
    for case in golden_cases["decode_cases"]:
        if case["shape"][0] * case["shape"][1] > 5000:
            continue
        post = decode_case_input(case, golden_decode)
        score, path = oracle_np.viterbi_py(post, case["klen"], case["skip_pen"], case["log"], case["nbase"])
        assert list(golden_decode["path_" + case["name"]]) == path, case["name"]
        assert float(score) == float.fromhex(case["score_hex"]), case["name"]


def test_viterbi_batch_layout(oracle, golden_decode):
    post = golden_decode["post_d50"]
    lp = np.log(post + 1e-10)
    rs = np.random.RandomState(0)
    other = np.log(rs.dirichlet(np.ones(1025) * 0.05, size=50).astype(np.float32) + 1e-10)
    lpb = np.ascontiguousarray(np.stack([lp, other], axis=1))
    scores, paths, lens = oracle.viterbi_batch(lpb, 5, skip_pen=3.0)
    s0, p0 = oracle.viterbi(lp, 5, skip_pen=3.0, log=True)
    s1, p1 = oracle.viterbi(other, 5, skip_pen=3.0, log=True)
    assert list(paths[0, :lens[0]]) == p0 and list(paths[1, :lens[1]]) == p1
    assert scores[0] == s0 and scores[1] == s1
    assert list(golden_decode["path_d50_skip3"]) == p0


def test_viterbi_rejects_short_kmers(oracle):
    with pytest.raises(AssertionError):
        oracle.viterbi(np.ones((4, 17)), 2)          # decode.py:50


def test_prepare_post(oracle, golden_prepare_post):
    g = golden_prepare_post
    assert np.array_equal(oracle.prepare_post(g["pp_in"], 1e-5), g["pp_out"])
    assert np.array_equal(oracle.prepare_post(g["pp_in"], 1e-3), g["pp_out_1e3"])
    # basecall.decode_post = prepare_post + viterbi   (basecall.py:26-51)
    for skip in (0.0, 5.0):
        post = oracle.prepare_post(g["dp_in"], 1e-5)
        score, call = oracle.viterbi(post, 5, skip_pen=skip)
        assert call == list(g["dp_call_skip%g" % skip])
        assert float(score) == float(g["dp_score_skip%g" % skip])
