"""GPU parity: decode.viterbi / prepare_post / argmax through the C ABI -- bit exact (integer paths, float32
scores) against the reference's known answers, the reference-generated goldens and the oracle."""
import numpy as np
import pytest

from tests.conftest import decode_case_input
from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu


def test_viterbi_reference_kats_float32(oracle, golden_decode):
    """test/unit/test_decode.py:233-256 (the reference runs them in float64; paths are equal in float32)."""
    need_gpu()
    from sloika_amd import decode
    score, path = decode.viterbi(golden_decode["kat_post3"], 3)
    assert path == [49, 7, 63, 63] and score == pytest.approx(-11.130084569094556, abs=1e-5)
    score, path = decode.viterbi(golden_decode["kat_post3"], 3, skip_pen=3.0)
    assert path == [49, 7, 31, 63, 63] and score == pytest.approx(-11.936803444063674, abs=1e-5)
    score, path = decode.viterbi(golden_decode["kat_mod_post"], 3, skip_pen=5.0, nbase=5)
    assert path == [int(x) - 1 for x in golden_decode["kat_mod_seq"] if x]


def test_viterbi_goldens_bit_exact_on_log_posteriors(oracle, golden_cases, golden_decode):
    """Same float32 log-posteriors as the reference saw -> identical path and identical float32 score."""
    need_gpu()
    from sloika_amd import decode
    for case in golden_cases["decode_cases"]:
        if case["dtype"] != "float32":
            continue
        post = decode_case_input(case, golden_decode)
        lpost = post if case["log"] else np.log(post + 1e-10)          # decode.py:56, numpy float32
        score, path = decode.viterbi(lpost, case["klen"], skip_pen=case["skip_pen"], log=True, nbase=case["nbase"])
        assert path == list(golden_decode["path_" + case["name"]]), case["name"]
        assert float(score) == float.fromhex(case["score_hex"]), case["name"]


def test_viterbi_post_modes_match_oracle_on_device_log(oracle, golden_decode):
    """post -> log on the device (SLK_POST_PLAIN / SLK_POST_RAW): the decoder must be bit-identical to the oracle
    run on the device's own log-posteriors, and those must be within 2 ulp-ish of numpy's."""
    torch = need_gpu()
    from sloika_amd import _lib, decode
    post = golden_decode["post_d50"]
    pd = dev(post)
    for mode, min_prob in ((_lib.POST_PLAIN, None), (_lib.POST_RAW, 1e-5)):
        lp = torch.empty_like(pd)
        _lib.check(_lib.lib().slk_log_post_f32(pd.data_ptr(), lp.data_ptr(), pd.numel(), mode,
                                               float(min_prob or 0.0), stream()))
        lph = lp.cpu().numpy()
        ref_in = post if min_prob is None else oracle.prepare_post(post[:, None, :], min_prob)
        np.testing.assert_allclose(lph, np.log(ref_in + np.float32(1e-10)), rtol=1e-6, atol=2e-6)
        for skip in (0.0, 3.0):
            scores, paths, lens = decode.viterbi_batch(pd[:, None, :], 5, skip_pen=skip, min_prob=min_prob)
            o_score, o_path = oracle.viterbi(lph, 5, skip_pen=skip, log=True)
            n = int(lens[0])
            assert paths[0, :n].cpu().tolist() == o_path and float(scores[0]) == float(o_score)
            assert (paths[0, n:] == -1).all()


@pytest.mark.parametrize("T,B,klen,nbase,skip", [(1, 3, 3, 4, 0.0), (2, 3, 3, 4, 1.0), (37, 5, 3, 4, 0.0), (64, 4, 4, 4, 2.5),
                                                 (65, 3, 5, 4, 0.0), (130, 7, 5, 4, 5.0), (40, 3, 3, 5, 1.0), (20, 2, 4, 5, 0.0),
                                                 (9, 2, 6, 4, 0.5)])
def test_viterbi_batch_vs_oracle(oracle, T, B, klen, nbase, skip):
    need_gpu()
    from sloika_amd import decode
    nst = nbase ** klen + 1
    rs = np.random.RandomState(T * 7 + B + klen)
    lp = np.log(rs.dirichlet(np.ones(nst) * 0.3, size=(T, B)).astype(np.float32) + np.float32(1e-6))
    if T > 4:
        lp[T // 2] = lp[T // 2, :, :1]                 # a fully tied row: every comparison at that step ties
    scores, paths, lens = decode.viterbi_batch(lp, klen, skip_pen=skip, log=True, nbase=nbase)
    o_scores, o_paths, o_lens = oracle.viterbi_batch(lp, klen, skip_pen=skip, nbase=nbase)
    assert np.array_equal(lens.cpu().numpy(), o_lens)
    assert np.array_equal(paths.cpu().numpy(), o_paths)
    assert np.array_equal(scores.cpu().numpy(), o_scores)


@pytest.mark.parametrize("klen,skip", [(5, 0.0), (4, 2.0), (3, 0.5)])
def test_viterbi_large_batch_kernel_equals_small_batch_kernel(oracle, klen, skip):
    """Batches of >= 1536 chunks take the wave-per-chunk forward kernel, smaller ones the workgroup-per-chunk kernel:
    same chunks through both must agree bit for bit (and with the oracle on a sample), ties included."""
    torch = need_gpu()
    from sloika_amd import decode
    T, B, nst = 33, 1600, 4 ** klen + 1
    g = torch.Generator(device="cuda").manual_seed(klen)
    lp = torch.log_softmax(3.0 * torch.randn((T, B, nst), device="cuda", generator=g), dim=2)
    lp[T // 2] = lp[T // 2, :, :1]                     # a fully tied row
    lp[5, :, 1:] = lp[5, :, 1:2]                       # all k-mers tied, blank different
    s_all, p_all, l_all = decode.viterbi_batch(lp, klen, skip_pen=skip, log=True)
    for lo, hi in ((0, 800), (800, 1600)):
        s_h, p_h, l_h = decode.viterbi_batch(lp[:, lo:hi].contiguous(), klen, skip_pen=skip, log=True)
        assert torch.equal(p_all[lo:hi], p_h) and torch.equal(l_all[lo:hi], l_h) and torch.equal(s_all[lo:hi], s_h)
    sub = [0, 1, 799, 1599]
    o_s, o_p, o_l = oracle.viterbi_batch(lp[:, sub].cpu().numpy(), klen, skip_pen=skip)
    assert np.array_equal(p_all[sub].cpu().numpy(), o_p) and np.array_equal(l_all[sub].cpu().numpy(), o_l)
    assert np.array_equal(s_all[sub].cpu().numpy(), o_s)


def test_viterbi_full_size_properties():
    """BASELINE size (T'=800, 1025 states): size-independent properties instead of a CPU re-run of every chunk:
    (1) replicating a chunk across the batch gives identical results in every slot, (2) the returned score
    equals the score of the returned path re-evaluated step by step, (3) -1 padding beyond len."""
    torch = need_gpu()
    from sloika_amd import decode
    rs = np.random.RandomState(0)
    T, B = 800, 16
    one = np.log(rs.dirichlet(np.ones(1025) * 0.05, size=T).astype(np.float32) + np.float32(1e-10))
    lp = np.repeat(one[:, None, :], B, axis=1)
    scores, paths, lens = decode.viterbi_batch(lp, 5, skip_pen=0.0, log=True)
    scores, paths, lens = scores.cpu().numpy(), paths.cpu().numpy(), lens.cpu().numpy()
    assert (scores == scores[0]).all() and (lens == lens[0]).all() and (paths == paths[0]).all()
    n = lens[0]
    assert 1 <= n <= T and (paths[0, n:] == -1).all() and (paths[0, :n] >= 0).all() and (paths[0, :n] < 1024).all()
    # consecutive path states must be step (shift by 1 base) or skip (shift by 2 bases) compatible
    p = paths[0, :n]
    ok_step = (p[1:] // 4) == (p[:-1] % 256)
    ok_skip = (p[1:] // 16) == (p[:-1] % 64)
    assert (ok_step | ok_skip).all()


def test_prepare_post_and_argmax(oracle, golden_prepare_post, golden_decode):
    need_gpu()
    from sloika_amd import decode
    g = golden_prepare_post
    assert np.array_equal(decode.prepare_post(g["pp_in"], 1e-5), g["pp_out"])
    assert np.array_equal(decode.prepare_post(g["pp_in"], 1e-3), g["pp_out_1e3"])
    with pytest.raises(ValueError):
        decode.prepare_post(np.zeros((4, 2, 5), dtype=np.float32))          # batch > 1: decode.py:30
    # test/unit/test_decode.py:201-204
    bases = decode.argmax(golden_decode["kat_post"].astype(np.float32), zero_is_blank=False)
    assert np.array_equiv(bases, golden_decode["kat_bases"])
    bases0 = decode.argmax(golden_decode["kat_post"].astype(np.float32), zero_is_blank=True)
    am = np.argmax(golden_decode["kat_post"], axis=1)
    assert np.array_equal(bases0, am[am != 0] - 1)


def test_decode_post_matches_reference_goldens(golden_prepare_post):
    """basecall.decode_post (sloika/basecall.py:26-51) on a raw posterior: path equal to the reference's."""
    need_gpu()
    from sloika_amd import basecall
    g = golden_prepare_post
    for skip in (0.0, 5.0):
        score, call = basecall.decode_post(g["dp_in"], 5, True, True, 1e-5, skip=skip)
        assert call == list(g["dp_call_skip%g" % skip])
        assert float(score) == pytest.approx(float(g["dp_score_skip%g" % skip]), rel=1e-6)


def test_viterbi_argument_errors():
    need_gpu()
    from sloika_amd import decode
    with pytest.raises(ValueError):
        decode.viterbi(np.ones((4, 17), dtype=np.float32), 2)              # decode.py:50
    with pytest.raises(ValueError):
        decode.viterbi(np.ones((4, 66), dtype=np.float32), 3)              # decode.py:52 nstate mismatch


def test_viterbi_on_logits_equals_viterbi_on_posterior(oracle):
    """Softmax + prepare_post + log + Viterbi in one pass over the logits == the same stages run separately,
    bit for bit, and == the oracle decoder on the log-posterior the device derives from the logits.
    Both the fused projection+statistics kernel (padded row stride) and the stand-alone statistics pass are covered."""
    torch = need_gpu()
    from sloika_amd import _lib, decode
    rs = np.random.RandomState(5)
    T, B, S, K = 90, 6, 1025, 96
    x = rs.normal(size=(T * B, K)).astype(np.float32)
    W = (rs.normal(size=(S, K)) * 0.6).astype(np.float32)
    b = rs.normal(size=S).astype(np.float32)
    b[0] += 3.0
    x[10 * B:11 * B] = 0.0                 # rows whose logits are just the bias
    xd, Wd, bd = dev(x), dev(W), dev(b)
    L = _lib.lib()
    for ld in (1056, 1025):
        logits = torch.full((T * B, ld), np.nan, dtype=torch.float32, device="cuda")
        stats = torch.empty((T * B, 2), dtype=torch.float32, device="cuda")
        assert L.slk_linear_rowstats_f32(xd.data_ptr(), K, Wd.data_ptr(), bd.data_ptr(), logits.data_ptr(), ld, T * B, K, S,
                                         stats.data_ptr(), stream()) == 0
        lg = logits[:, :S].cpu().numpy()
        ref_l = x.astype(np.float64) @ W.astype(np.float64).T + b
        np.testing.assert_allclose(lg, ref_l, atol=2e-5)
        np.testing.assert_array_equal(stats[:, 0].cpu().numpy(), lg.max(axis=1))
        np.testing.assert_allclose(1.0 / stats[:, 1].cpu().numpy(),
                                   np.exp(lg - lg.max(axis=1, keepdims=True)).astype(np.float64).sum(axis=1), rtol=2e-6)
        post = torch.empty((T, B, S), dtype=torch.float32, device="cuda")
        assert L.slk_softmax_from_stats_f32(logits.data_ptr(), ld, stats.data_ptr(), post.data_ptr(), S, T * B, S, stream()) == 0
        np.testing.assert_allclose(post.cpu().numpy().sum(axis=2), 1.0, atol=1e-5)
        lp = torch.empty((T, B, S), dtype=torch.float32, device="cuda")
        assert L.slk_log_post_logits_f32(logits.data_ptr(), ld, stats.data_ptr(), lp.data_ptr(), T * B, S, 1e-5, stream()) == 0
        for skip in (0.0, 4.0):
            s1, p1, l1 = decode.viterbi_logits_batch(logits, stats, 5, T, B, ld=ld, skip_pen=skip, min_prob=1e-5)
            s2, p2, l2 = decode.viterbi_batch(post, 5, skip_pen=skip, min_prob=1e-5)
            assert torch.equal(p1, p2) and torch.equal(l1, l2) and torch.equal(s1, s2)
            o_s, o_p, o_l = oracle.viterbi_batch(lp.cpu().numpy(), 5, skip_pen=skip)
            assert np.array_equal(p1.cpu().numpy(), o_p) and np.array_equal(l1.cpu().numpy(), o_l)
            assert np.array_equal(s1.cpu().numpy(), o_s)
        ref = np.log(np.float32(1e-5) + np.float32(1 - 1e-5) * post.cpu().numpy() + np.float32(1e-10))
        np.testing.assert_allclose(lp.cpu().numpy(), ref, rtol=1e-6, atol=2e-6)
    # stand-alone statistics pass over dense logits (the K > 128 fallback) agrees with the in-place softmax kernel
    dense = logits[:, :S].contiguous()
    st2 = torch.empty((T * B, 2), dtype=torch.float32, device="cuda")
    assert L.slk_softmax_rowstats_f32(dense.data_ptr(), T * B, S, st2.data_ptr(), stream()) == 0
    p_a = torch.empty((T * B, S), dtype=torch.float32, device="cuda")
    assert L.slk_softmax_from_stats_f32(dense.data_ptr(), S, st2.data_ptr(), p_a.data_ptr(), S, T * B, S, stream()) == 0
    p_b = dense.clone()
    assert L.slk_softmax_rows_f32(p_b.data_ptr(), T * B, S, stream()) == 0
    assert torch.equal(p_a, p_b)
