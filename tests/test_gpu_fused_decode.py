"""csrc/softmax_viterbi.hip: the Softmax layer's projection, softmax (sloika/layers.py:309-314), prepare_post
(decode.py:21-36), log and the Viterbi forward pass (decode.py:39-82) in ONE kernel -- the logits never exist in memory.

What is checked, all through the C ABI (slk_softmax_viterbi_f32):
  * the log-posteriors the kernel's dynamic programme consumed (its lp_dump output) against a float64 evaluation of the
    reference formulas (tolerance written below),
  * paths, lengths and float32 scores against the ORACLE decoder run on exactly those log-posteriors: bit for bit (integer /
    index work; max-plus float32 arithmetic is order independent),
  * agreement with the two-kernel path (projection kernel + decoder on the logits).
"""
import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev

pytestmark = pytest.mark.gpu
LP_TOL = 2e-5          # log-posteriors, absolute (they are logs of numbers >= 1e-5: 2e-5 relative on the posterior)
POST_TOL = 2e-6        # posteriors min_prob + (1 - min_prob) p, absolute


def _reference_lp(x, W, b, min_prob=1e-5):
    l64 = x.astype(np.float64) @ W.astype(np.float64).T + b.astype(np.float64)
    l64 -= l64.max(axis=2, keepdims=True)                              # layers.py:311
    p = np.exp(l64)
    p /= p.sum(axis=2, keepdims=True)                                  # layers.py:312-313
    return np.log(min_prob + (1.0 - min_prob) * p + 1e-10)            # decode.py:36, :56


def _softmax_layer(rs, K, S=1025, wscale=0.5):
    from sloika_amd import layers
    W = (rs.normal(size=(S, K)) * wscale).astype(np.float32)
    b = rs.normal(size=S).astype(np.float32)
    b[0] += 3.0
    sm = layers.Softmax(K, S, has_bias=True)
    sm.W.set_value(W)
    sm.b.set_value(b)
    return sm, W, b


def _check(oracle, rs, T, B, K, skip=0.0, ragged=False, wscale=0.5, xscale=1.0, lp_tol=LP_TOL, post_tol=POST_TOL):
    torch = need_gpu()
    from sloika_amd import decode
    sm, W, b = _softmax_layer(rs, K, wscale=wscale)
    pack = sm.viterbi_pack(4, 5)
    assert pack is not None
    x = (np.tanh(rs.normal(size=(T, B, K))) * xscale).astype(np.float32)
    x[min(3, T - 1)] = 0.0                                             # rows whose logits are just the bias
    xd = dev(x)
    ln = np.full(B, T, dtype=np.int32)
    lens = None
    if ragged:
        ln = rs.randint(1, T + 1, size=B).astype(np.int32)
        ln[0] = T
        lens = torch.from_numpy(ln).cuda()
    dump = torch.full((T, B, 1025), float("nan"), dtype=torch.float32, device="cuda")
    sc, pa, le = decode.viterbi_fused_batch(xd, pack, 5, skip_pen=skip, lengths=lens, lp_dump=dump)
    sc0, pa0, le0 = decode.viterbi_fused_batch(xd, pack, 5, skip_pen=skip, lengths=lens)      # the kernel without the dump
    assert torch.equal(sc, sc0) and torch.equal(pa, pa0) and torch.equal(le, le0)
    lp = dump.cpu().numpy()
    ref = _reference_lp(x, W, b)
    scn, pan, len_ = sc.cpu().numpy(), pa.cpu().numpy(), le.cpu().numpy()
    for bb in range(B):
        Tb = int(ln[bb])
        assert np.abs(lp[:Tb, bb] - ref[:Tb, bb]).max() < lp_tol
        assert np.abs(np.exp(lp[:Tb, bb]) - np.exp(ref[:Tb, bb])).max() < post_tol
        o_s, o_p, o_l = oracle.viterbi_batch(np.ascontiguousarray(lp[:Tb, bb:bb + 1]), 5, skip_pen=skip)
        assert o_l[0] == len_[bb]
        assert np.array_equal(o_p[0, :o_l[0]], pan[bb, :len_[bb]]) and (pan[bb, len_[bb]:] == -1).all()
        assert o_s[0] == scn[bb]                                       # float32 score, bit for bit
    # the two-kernel path decodes the same sequences (its log-posteriors differ in the last bits, so a near tie could in
    # principle resolve the other way: scores must agree to float32 accumulation accuracy, paths on all but such ties)
    logits, stats, ld = sm.logits_and_stats(xd)
    s2, p2, l2 = decode.viterbi_logits_batch(logits, stats, 5, T, B, ld=ld, skip_pen=skip, lengths=lens)
    np.testing.assert_allclose(scn, s2.cpu().numpy(), rtol=2e-6, atol=1e-4)
    assert (pa == p2).all(dim=1).float().mean().item() >= 0.75


@pytest.mark.parametrize("T,B,K,skip,ragged", [
    (50, 5, 96, 0.0, False),          # odd batch: the last workgroup holds one chunk
    (37, 2, 96, 4.0, False),          # skip penalty, a partial last block of 16 steps
    (1, 3, 96, 0.0, False),           # T = 1: only the initialisation step (decode.py:57)
    (16, 1, 64, 0.0, False),          # exactly one block, a single chunk
    (33, 4, 128, 0.0, False),
    (70, 7, 112, 0.0, True),          # ragged batch (whole reads of different lengths)
    (200, 6, 96, 0.0, True),
    (9, 9, 96, 5.0, True),
    (2, 2, 64, 0.0, False),
])
def test_fused_decode_against_oracle_on_its_log_posteriors(oracle, T, B, K, skip, ragged):
    _check(oracle, np.random.RandomState(1000 * T + B), T, B, K, skip, ragged)


@pytest.mark.parametrize("T", [15, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 66, 95, 97, 129, 161])
def test_fused_decode_backtrace_block_edges(oracle, T):
    """viterbi_backtrace_rows_kernel (csrc/decode.hip) walks sixteen rows per register block, two blocks per loop iteration, and
    stores the path in windows of 64 entries: chunk lengths on either side of every one of those boundaries, full and ragged,
    six chunks (one and a half workgroups of four waves)."""
    _check(oracle, np.random.RandomState(7000 + T), T, 6, 64, 0.0, False)
    _check(oracle, np.random.RandomState(8000 + T), T, 6, 64, 2.0, True)


def test_fused_decode_operand_ranges(oracle):
    """Row scaling of x and column scaling of W: large weights (the trained pickle reaches 6), tiny and huge activations."""
    # logits of magnitude ~50: one float32 ulp of the logit is 4e-6, and so is the posterior's absolute error
    _check(oracle, np.random.RandomState(5), 40, 4, 96, wscale=3.0, lp_tol=1e-4, post_tol=1e-5)
    _check(oracle, np.random.RandomState(6), 40, 4, 112, wscale=0.5, xscale=1e-3)
    _check(oracle, np.random.RandomState(7), 40, 4, 64, wscale=0.02, xscale=50.0)


def test_fused_decode_unsupported_shapes_fall_back():
    """Shapes the kernel does not cover are refused by the C ABI (no silent approximation) and the layer reports it."""
    torch = need_gpu()
    from sloika_amd import _lib, layers
    L = _lib.lib()
    assert L.slk_softmax_viterbi_pack_bytes(96, 4, 5) > 0 and L.slk_softmax_viterbi_pack_bytes(128, 4, 5) > 0
    assert L.slk_softmax_viterbi_pack_bytes(80, 4, 5) == 0             # insize not one of 64, 96, 112, 128
    assert L.slk_softmax_viterbi_pack_bytes(96, 4, 4) == 0 and L.slk_softmax_viterbi_pack_bytes(96, 5, 5) == 0
    assert layers.Softmax(80, 1025).viterbi_pack(4, 5) is None
    assert layers.Softmax(96, 257).viterbi_pack(4, 4) is None


def test_fused_decode_batch_composition(oracle):
    """A chunk's result does not depend on its neighbours or its slot (what makes chunk sharding exact): replicated chunks give
    identical rows, and a sub-batch decodes like the same chunks inside the big batch."""
    torch = need_gpu()
    from sloika_amd import decode
    rs = np.random.RandomState(11)
    sm, W, b = _softmax_layer(rs, 96)
    pack = sm.viterbi_pack(4, 5)
    one = np.tanh(rs.normal(size=(120, 1, 96))).astype(np.float32)
    x = np.concatenate([np.repeat(one, 5, axis=1), np.tanh(rs.normal(size=(120, 6, 96))).astype(np.float32)], axis=1)
    sc, pa, le = decode.viterbi_fused_batch(dev(x), pack, 5)
    assert (pa[:5] == pa[0]).all() and (sc[:5] == sc[0]).all() and (le[:5] == le[0]).all()
    s2, p2, l2 = decode.viterbi_fused_batch(dev(np.ascontiguousarray(x[:, 4:9])), pack, 5)
    assert torch.equal(p2, pa[4:9]) and torch.equal(s2, sc[4:9]) and torch.equal(l2, le[4:9])


def test_fused_decode_full_size_sampled_chunks(oracle):
    """BASELINE.json configs[2] size (T' = 800, batch 1024, insize 96): eight chunks picked at random, oracle decoder on the
    dumped log-posteriors of those chunks, bit for bit; the dump itself against float64."""
    torch = need_gpu()
    from sloika_amd import decode
    rs = np.random.RandomState(12)
    T, B, K = 800, 1024, 96
    sm, W, b = _softmax_layer(rs, K)
    pack = sm.viterbi_pack(4, 5)
    xd = torch.tanh(torch.randn((T, B, K), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
    dump = torch.empty((T, B, 1025), dtype=torch.float32, device="cuda")
    sc, pa, le = decode.viterbi_fused_batch(xd, pack, 5, lp_dump=dump)
    sc0, pa0, le0 = decode.viterbi_fused_batch(xd, pack, 5)
    assert torch.equal(sc, sc0) and torch.equal(pa, pa0) and torch.equal(le, le0)
    pick = np.sort(rs.choice(B, size=8, replace=False))
    idx = torch.from_numpy(pick).cuda()
    lp = dump[:, idx, :].contiguous().cpu().numpy()
    ref = _reference_lp(xd[:, idx, :].cpu().numpy(), W, b)
    assert np.abs(lp - ref).max() < LP_TOL
    o_s, o_p, o_l = oracle.viterbi_batch(lp, 5, skip_pen=0.0)
    assert np.array_equal(le.cpu().numpy()[pick], o_l)
    assert np.array_equal(pa.cpu().numpy()[pick], o_p)
    assert np.array_equal(sc.cpu().numpy()[pick], o_s)


def test_fused_decode_repeats_bit_for_bit():
    """Race / hazard screen: the same call repeated gives the same bits, with and without the dump (two instantiations of the
    kernel), at the insize where a destination/operand register overlap of v_mfma_f32_32x32x16_f16 once corrupted a row in
    some runs (csrc/softmax_viterbi.hip, mma_pair), and at full size."""
    torch = need_gpu()
    from sloika_amd import decode
    for (T, B, K, reps) in ((40, 4, 64, 12), (40, 5, 96, 6), (40, 4, 112, 6), (40, 4, 128, 6), (800, 1024, 64, 3), (800, 1024, 96, 3)):
        rs = np.random.RandomState(K + T)
        sm, W, b = _softmax_layer(rs, K)
        pack = sm.viterbi_pack(4, 5)
        xd = torch.tanh(torch.randn((T, B, K), device="cuda", generator=torch.Generator(device="cuda").manual_seed(K)))
        dump = torch.empty((T, B, 1025), dtype=torch.float32, device="cuda")
        first = None
        for rep in range(reps):
            sc, pa, le = decode.viterbi_fused_batch(xd, pack, 5, lp_dump=dump if rep % 2 else None)
            if first is None:
                first = (sc.clone(), pa.clone(), le.clone())
            else:
                assert torch.equal(sc, first[0]) and torch.equal(pa, first[1]) and torch.equal(le, first[2]), (T, B, K, rep)
