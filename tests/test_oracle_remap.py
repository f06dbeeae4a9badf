"""The remap -> mapping table -> chunk labels oracle (oracle/oracle_remap.py) against the reference's own outputs
(tests/golden/remap.npz, made by tests/golden/make_remap_goldens.py from sloika/tools/chunkify_raw.py:260-296, 164-210),
and the host bookkeeping of sloika_amd.chunkify_raw (no GPU needed)."""
import hashlib
import os
import sys

import numpy as np
import pytest

from oracle import oracle, oracle_remap

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
import make_remap_goldens as mrg          # noqa: E402  (input generators only; nothing of the reference is imported)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "remap.npz"))


def case(gold, ci):
    tag = "c%d_" % ci
    seed, nsamp = (int(v) for v in gold[tag + "seed_nsamp"])
    ref = gold[tag + "ref"].tobytes()
    signal, states, post = mrg.case_inputs(seed, nsamp, ref)
    assert _sha(signal) == str(gold[tag + "signal_sha256"]) and _sha(post) == str(gold[tag + "post_sha256"])
    prior = tuple(None if p < 0 else float(p) for p in gold["cases_prior"][ci])
    return dict(tag=tag, ref=ref, signal=signal, states=states, post=post, prior=prior, slip=float(gold["cases_slip"][ci]),
                chunk_len=int(gold["cases_chunk_len"][ci]), ds=int(gold["cases_downsample"][ci]),
                norm=str(gold["cases_normalisation"][ci]))


NCASE = 3


@pytest.mark.parametrize("ci", range(NCASE))
def test_oracle_raw_remap_matches_reference(gold, ci):
    c = case(gold, ci)
    tag = c["tag"]
    score, cols, path, seq = oracle_remap.raw_remap(c["ref"], c["signal"], c["post"], 1e-5, 5, c["prior"], c["slip"])
    assert list(seq) == c["states"] == list(gold[tag + "seq"])
    assert np.array_equal(path, gold[tag + "path"])
    assert np.float32(score) == gold[tag + "score"]
    for f in ("start", "length", "seq_pos", "move"):
        assert np.array_equal(cols[f], gold[tag + "mt_" + f]), f
    kmers = np.array([c["ref"][i:i + 5] for i in range(len(c["ref"]) - 4)])
    assert np.array_equal(kmers[cols["seq_pos"]], gold[tag + "mt_kmer"])
    assert gold[tag + "mt_good"].all()


@pytest.mark.parametrize("ci", range(NCASE))
def test_oracle_chunk_labels_match_reference(gold, ci):
    c = case(gold, ci)
    tag = c["tag"]
    cols = {f: gold[tag + "mt_" + f].astype(np.int64) for f in ("start", "length", "seq_pos", "move")}
    nsamp = len(c["signal"])
    ml = nsamp // c["chunk_len"]
    trimmed = oracle_remap.trim_table(cols, nsamp, 0, ml * c["chunk_len"])
    states = oracle_remap.states_of_reference(c["ref"], 5)[trimmed["seq_pos"]]
    plain = oracle_remap.chunk_labels(trimmed, states, ml, c["chunk_len"], c["ds"])
    assert plain.dtype == np.int32 and str(gold[tag + "plain_labels_dtype"]) == "int32"
    assert np.array_equal(plain, gold[tag + "plain_labels"])
    interp = oracle_remap.chunk_labels_interp(trimmed, c["ref"], ml, c["chunk_len"], c["ds"], 5, 5)
    assert np.array_equal(interp, gold[tag + "interp_labels"])
    assert not gold[tag + "plain_bad"].any() and gold[tag + "plain_bad"].shape == (ml, c["chunk_len"])
    # the chunks are the normalised signal (oracle.med_mad_normalise is pinned by test_oracle_signal.py)
    block = c["signal"][: ml * c["chunk_len"]]
    if c["norm"] == "per-chunk":
        want = oracle.med_mad_normalise(block.reshape(ml, -1))
    elif c["norm"] == "per-read":
        want = oracle.med_mad_normalise(block.reshape(1, -1)).reshape(ml, -1)
    else:
        want = block.reshape(ml, -1)
    assert np.array_equal(want, gold[tag + "plain_chunks"]) and np.array_equal(want, gold[tag + "interp_chunks"])


# ---- host bookkeeping of the product module (record arrays; no device work) ------------------------------------------------

def _table(gold, tag):
    n = len(gold[tag + "mt_start"])
    t = np.zeros(n, dtype=[("start", "<i8"), ("length", "<i8"), ("seq_pos", "<i8"), ("move", "<i8"), ("kmer", "S5"),
                           ("good_emission", "?")])
    for f in ("start", "length", "seq_pos", "move", "kmer"):
        t[f] = gold[tag + "mt_" + f]
    t["good_emission"] = gold[tag + "mt_good"]
    return t


def test_trim_and_registration_follow_the_oracle(gold):
    from sloika_amd import chunkify_raw as cr
    c = case(gold, 1)
    t = _table(gold, c["tag"])
    assert cr.mapping_table_is_registered(c["signal"], t)
    sig, tt = cr.trim_signal_and_mapping(c["signal"], t, 0, 4000)
    cols = {f: t[f].astype(np.int64) for f in ("start", "length", "seq_pos", "move")}
    want = oracle_remap.trim_table(cols, len(c["signal"]), 0, 4000)
    assert len(sig) == 4000 and cr.mapping_table_is_registered(sig, tt)
    for f in want:
        assert np.array_equal(tt[f], want[f])
    sig, tt = cr.trim_signal_and_mapping(c["signal"], t, 123, 3210)
    want = oracle_remap.trim_table(cols, len(c["signal"]), 123, 3210)
    assert len(sig) == 3210 - 123 and cr.mapping_table_is_registered(sig, tt)
    for f in want:
        assert np.array_equal(tt[f], want[f])
    broken = t.copy()
    broken["length"][5] += 1
    assert not cr.mapping_table_is_registered(c["signal"], broken)
    assert not cr.mapping_table_is_registered(c["signal"][:-1], t)


def test_convert_mapping_times_to_samples():
    """The case of the reference's test/unit/test_raw_chunkify_utils.py: 5 events over 49 samples at 4 kHz."""
    from sloika_amd import chunkify_raw as cr
    sample_rate, start_sample = 4000.0, 10000
    bounds = np.array([0, 9, 20, 28, 41, 49])
    t = np.zeros(5, dtype=[("start", "<f8"), ("length", "<f8"), ("kmer", "S5")])
    t["start"] = (bounds[:-1] + start_sample) / sample_rate
    t["length"] = np.diff(bounds) / sample_rate
    out = cr.convert_mapping_times_to_samples(t, start_sample, sample_rate)
    assert out["start"].dtype == np.int64 and out["length"].dtype == np.int64
    assert np.array_equal(out["start"], bounds[:-1]) and np.array_equal(out["length"], np.diff(bounds))
    assert out.dtype.names == t.dtype.names


def test_small_helpers():
    from sloika_amd import chunkify_raw as cr
    assert list(cr.replace_repeats_with_zero(np.array([3, 3, 4, 4, 4, 1]))) == [3, 0, 4, 0, 0, 1]
    assert list(cr.fill_zeros_with_prev(np.array([0, 2, 0, 0, 5, 0]))) == [0, 2, 2, 2, 5, 5]
    assert list(cr.index_of_previous_non_zero(np.array([1, 0, 0, 2, 0, -1, 1]))) == [0, 0, 0, 3, 3, 3, 6]
