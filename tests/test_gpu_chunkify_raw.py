"""raw_remap and raw_chunkify of sloika_amd.chunkify_raw (C ABI: csrc/transducer.hip, chunk_labels.hip, frontend.hip) against
the reference's own outputs (tests/golden/remap.npz) and against the oracle on generated tables.  Bit-for-bit: scores, paths,
mapping tables, labels, bad flags and the normalised chunks."""
import numpy as np
import pytest

from oracle import oracle_remap
from tests.test_oracle_remap import GOLD, NCASE, _table, case, gold     # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _stub_network(post):
    """The compiled model of the fixture: returns the stored posterior for the read (as make_remap_goldens.py did)."""
    def calc_post(inmat):
        import torch
        assert inmat.shape[1:] == (1, 1)
        return torch.from_numpy(post[:, None, :]).to(inmat.device) if hasattr(inmat, "device") else post[:, None, :]
    return calc_post


def _check_table(table, gold, tag):
    for f in ("start", "length", "seq_pos", "move"):
        assert table[f].dtype == np.int64 and np.array_equal(table[f], gold[tag + "mt_" + f]), f
    assert np.array_equal(table["kmer"], gold[tag + "mt_kmer"]) and table["good_emission"].all()


@pytest.mark.parametrize("ci", range(NCASE))
def test_raw_remap_matches_reference(gold, ci):
    from sloika_amd import batch, chunkify_raw as cr
    batch.init_chunk_identity_worker(5, b"ACGT")
    c = case(gold, ci)
    tag = c["tag"]
    score, table, path, seq = cr.raw_remap(c["ref"], c["signal"], 1e-5, 5, c["prior"], c["slip"],
                                           calc_post=_stub_network(c["post"]))
    assert np.float32(score) == gold[tag + "score"]
    assert path.dtype == np.int64 and np.array_equal(path, gold[tag + "path"])
    assert list(seq) == list(gold[tag + "seq"])
    _check_table(table, gold, tag)
    assert cr.mapping_table_is_registered(c["signal"], table)


def test_raw_remap_many_is_one_launch_of_the_same(gold):
    from sloika_amd import batch, chunkify_raw as cr
    batch.init_chunk_identity_worker(5, b"ACGT")
    cs = [case(gold, ci) for ci in range(NCASE)]
    posts = {len(c["signal"]): c["post"] for c in cs}

    def calc_post(inmat):
        import torch
        return torch.from_numpy(posts[inmat.shape[0]][:, None, :]).to(inmat.device)
    for prior, slip in (((25.0, 25.0), 5.0), ((None, None), 2.5)):
        many = cr.raw_remap_many([c["ref"] for c in cs], [c["signal"] for c in cs], 1e-5, 5, prior, slip, calc_post=calc_post)
        for c, (score, table, path, seq) in zip(cs, many):
            s1, t1, p1, q1 = cr.raw_remap(c["ref"], c["signal"], 1e-5, 5, prior, slip, calc_post=calc_post)
            assert np.float32(score) == np.float32(s1) and np.array_equal(path, p1) and list(seq) == list(q1)
            assert np.array_equal(table, t1)
            want = oracle_remap.raw_remap(c["ref"], c["signal"], c["post"], 1e-5, 5, prior, slip)
            # np.log of the posterior (transducer.py:30) is a float32 library call whose last bit differs between numpy
            # builds, glibc (the oracle) and the device: scores agree to float32 rounding, paths exactly
            assert float(score) == pytest.approx(float(want[0]), rel=2e-6) and np.array_equal(want[2], path)


@pytest.mark.parametrize("ci", range(NCASE))
@pytest.mark.parametrize("interp", [False, True])
def test_raw_chunkify_matches_reference(gold, ci, interp):
    from sloika_amd import batch, chunkify_raw as cr
    batch.init_chunk_identity_worker(5, b"ACGT")
    c = case(gold, ci)
    tag = c["tag"]
    attrs = {"reference": c["ref"], "direction": "+", "ref_start": 0}
    chunks, labels, bad = cr.raw_chunkify(c["signal"], _table(gold, tag), c["chunk_len"], 5, c["norm"], c["ds"], interp, attrs)
    k = tag + ("interp_" if interp else "plain_")
    # (the reference's interpolated labels are int64 -- np.array(...) + 1, chunkify_raw.py:113 -- the others 'i4', :205)
    assert labels.dtype == (np.int64 if interp else np.int32) and np.array_equal(labels, gold[k + "labels"])
    assert bad.dtype == bool and np.array_equal(bad, gold[k + "bad"])
    assert chunks.dtype == np.float32 and chunks.shape == gold[k + "chunks"].shape + (1,)
    assert np.array_equal(chunks[:, :, 0], gold[k + "chunks"])


def test_raw_chunk_worker_with_a_stored_mapping(gold, capsys):
    """chunkify_raw.py:213-257 (`chunkify raw_identity`): mapping times in seconds -> samples, trim, register, chunk.  The reference
    reads the stored mapping through untangled.fast5 (a dependency, not in its tree); here a stand-in object offers the same four
    things, built from a golden table shifted by a start time and expressed in seconds."""
    from sloika_amd import batch, chunkify_raw as cr
    batch.init_chunk_identity_worker(5, b"ACGT")
    c = case(gold, 0)
    tag = c["tag"]
    table = _table(gold, tag)
    rate, start_time, lead = 4000.0, 12000, 37                      # 37 unmapped samples in front of the mapped part
    secs = np.zeros(len(table), dtype=[(n, "<f8" if n in ("start", "length") else k) for n, k in table.dtype.descr])
    for n in table.dtype.names:
        secs[n] = table[n]
    secs["start"] = (table["start"] + lead + start_time) / rate
    secs["length"] = table["length"] / rate
    signal = np.concatenate([np.full(lead, 50.0, dtype=np.float32), c["signal"], np.full(11, 50.0, dtype=np.float32)])
    attrs = {"reference": c["ref"], "direction": "+", "ref_start": 0}

    class Stored(object):
        sample_rate = rate

        def __init__(self):
            self.start_time = start_time

        def get_any_mapping_data(self, section):
            assert section == "template"
            return secs, attrs

        def get_read(self, raw=True):
            return signal

    got = cr.raw_chunk_worker(Stored(), c["chunk_len"], 5, 0, (0, 0), c["norm"], c["ds"])
    want = cr.raw_chunkify(c["signal"], table, c["chunk_len"], 5, c["norm"], c["ds"], False)
    for a, b in zip(got, want):
        assert a.flags["C_CONTIGUOUS"] and a.dtype == b.dtype and np.array_equal(a, b)
    assert np.array_equal(got[1], gold[tag + "plain_labels"])
    # too short for min_length, and a file without a stored mapping: None and the reference's messages
    assert cr.raw_chunk_worker(Stored(), c["chunk_len"], 5, 10 ** 9, (0, 0), c["norm"], c["ds"]) is None
    assert "is too short" in capsys.readouterr().err
    assert cr.raw_chunk_worker("/nonexistent/read.fast5", 500, 5, 0, (0, 0), "per-read", 5) is None
    assert "Failed to get mapping data from /nonexistent/read.fast5" in capsys.readouterr().err


def test_remap_then_chunkify_end_to_end(gold):
    """The worker's chain (chunkify_raw.py:323-335): the table raw_remap returns goes straight into raw_chunkify."""
    from sloika_amd import batch, chunkify_raw as cr
    batch.init_chunk_identity_worker(5, b"ACGT")
    c = case(gold, 0)
    tag = c["tag"]
    _, table, _, _ = cr.raw_remap(c["ref"], c["signal"], 1e-5, 5, c["prior"], c["slip"], calc_post=_stub_network(c["post"]))
    chunks, labels, bad = cr.raw_chunkify(c["signal"], table, c["chunk_len"], 5, c["norm"], c["ds"], False)
    assert np.array_equal(labels, gold[tag + "plain_labels"]) and np.array_equal(chunks[:, :, 0], gold[tag + "plain_chunks"])
    assert not bad.any()


def _random_table(rs, nsample, nref, zero_lengths):
    """A registered mapping table with irregular block lengths (some empty), stays, skips and backward steps."""
    nblock = int(rs.randint(3, max(4, nsample // 6)))
    cuts = np.sort(rs.choice(np.arange(1, nsample), size=nblock - 1, replace=zero_lengths))    # repeats = empty blocks
    bounds = np.concatenate([[0], cuts, [nsample]])
    nblock = len(bounds) - 1
    step = rs.choice([0, 0, 1, 1, 1, 2, -1], size=nblock)
    step[0] = 0
    pos = np.clip(np.cumsum(step) + 2, 0, nref - 5)
    t = np.zeros(nblock, dtype=[("start", "<i8"), ("length", "<i8"), ("seq_pos", "<i8"), ("move", "<i8"), ("kmer", "S5"),
                                ("good_emission", "?")])
    t["start"], t["length"], t["seq_pos"] = bounds[:-1], np.diff(bounds), pos
    t["move"] = np.ediff1d(pos, to_begin=1)
    return t


@pytest.mark.parametrize("seed", range(6))
def test_raw_chunkify_many_against_oracle(seed):
    from sloika_amd import batch, chunkify_raw as cr
    batch.init_chunk_identity_worker(5, b"ACGT")
    rs = np.random.RandomState(700 + seed)
    chunk_len, ds = [(64, 5), (100, 1), (250, 7), (33, 4), (500, 5), (128, 128)][seed]
    signals, tables, refs = [], [], []
    for r in range(int(rs.randint(1, 9))):
        nsample = int(rs.randint(chunk_len, 12 * chunk_len))
        ref = bytes(rs.choice(list(b"ACGT"), size=int(rs.randint(40, 400))).tolist())
        t = _random_table(rs, nsample, len(ref), zero_lengths=bool(seed % 2))
        t["kmer"] = np.array([ref[i:i + 5] for i in range(len(ref) - 4)])[t["seq_pos"]]
        signals.append((rs.normal(size=nsample) * 10 + 80).astype(np.float32))
        tables.append(t)
        refs.append(ref)
    norm = ["per-chunk", "per-read", "none"][seed % 3]
    got = cr.raw_chunkify_many(signals, tables, chunk_len, 5, norm, ds)
    for sig, t, ref, (chunks, labels, bad) in zip(signals, tables, refs, got):
        ml = len(sig) // chunk_len
        cols = {f: t[f].astype(np.int64) for f in ("start", "length", "seq_pos", "move")}
        trimmed = oracle_remap.trim_table(cols, len(sig), 0, ml * chunk_len)
        states = oracle_remap.states_of_reference(ref, 5)[trimmed["seq_pos"]]
        want = oracle_remap.chunk_labels(trimmed, states, ml, chunk_len, ds)
        assert labels.shape == want.shape and np.array_equal(labels, want)
        assert bad.shape == (ml, chunk_len) and not bad.any() and chunks.shape == (ml, chunk_len, 1)
        one = cr.raw_chunkify(sig, t, chunk_len, 5, norm, ds, False)
        assert np.array_equal(one[1], labels) and np.array_equal(one[0], chunks)
        # interpolated labels of the same table
        attrs = {"reference": ref, "direction": "+", "ref_start": 0}
        if len(range(0, ml * chunk_len, ds)) % ml:
            with pytest.raises(ValueError):           # the reference's reshape((ml, -1)) fails the same way (:192)
                cr.raw_chunkify(sig, t, chunk_len, 5, norm, ds, True, attrs)
            continue
        want_i = oracle_remap.chunk_labels_interp(trimmed, ref, ml, chunk_len, ds, 5, 5)
        got_i = cr.raw_chunkify(sig, t, chunk_len, 5, norm, ds, True, attrs)[1]
        assert np.array_equal(got_i, want_i)


def test_interpolate_closures_and_label_lookup(gold):
    from sloika_amd import batch, chunkify_raw as cr
    batch.init_chunk_identity_worker(5, b"ACGT")
    c = case(gold, 2)
    t = _table(gold, c["tag"])
    cols = {f: t[f].astype(np.int64) for f in ("start", "length", "seq_pos", "move")}
    times = np.array([0.0, 3.5, 17.0, 1000.25, 4321.0, 9000.0, 20000.0])
    for forward, att in ((True, {"direction": "+", "ref_start": 3, "reference": c["ref"]}),
                         (False, {"direction": "-", "ref_stop": 480, "ref_start": 0, "reference": c["ref"]})):
        pos = cr.interpolate_pos(t, att)(times, 5)
        want = oracle_remap.interp_positions(cols, times, 5, 5, forward, att["ref_start"] if forward else att["ref_stop"])
        assert pos.dtype == np.int64 and np.array_equal(pos, want)
    att = {"direction": "+", "ref_start": 0, "reference": c["ref"]}
    lab = cr.interpolate_labels(t, att)(times, 5)
    want = oracle_remap.states_of_reference(c["ref"], 5)[oracle_remap.interp_positions(cols, times, 5, 5)]
    assert np.array_equal(lab, want)
    # labels_from_mapping_table: the middle 3 letters of 5-mers, any array shape
    batch.init_chunk_identity_worker(3, b"ACGT")
    km = t["kmer"][:12].reshape(3, 4)
    got = cr.labels_from_mapping_table(km, 3)
    want = np.array([oracle_remap.states_of_reference(k[1:4], 3)[0] for k in km.flat]).reshape(3, 4)
    assert got.dtype == np.int32 and np.array_equal(got, want)
    assert np.array_equal(cr.labels_from_mapping_table(km, 3, index_from=0), want - 1)
    batch.init_chunk_identity_worker(5, b"ACGT")
    with pytest.raises(KeyError):
        cr.labels_from_mapping_table(np.array([b"ACGTN"]), 5)


def test_errors_are_loud(gold):
    from sloika_amd import batch, chunkify_raw as cr
    batch.init_chunk_identity_worker(5, b"ACGT")
    c = case(gold, 1)
    t = _table(gold, c["tag"])
    with pytest.raises(AssertionError):
        cr.raw_chunkify(c["signal"][:100], t, 500, 5, "per-chunk", 5, False)          # shorter than a chunk
    with pytest.raises(AssertionError):
        cr.raw_chunkify(c["signal"][:-3], t, 500, 5, "per-chunk", 5, False)           # not registered
    with pytest.raises(AssertionError):
        cr.raw_chunkify(c["signal"], t, 500, 5, "per-banana", 5, False)
    saved, batch.calc_post = batch.calc_post, None
    try:
        with pytest.raises(ValueError):
            cr.raw_remap(c["ref"], c["signal"], 1e-5, 5, (None, None), 5.0)           # no compiled model anywhere
    finally:
        batch.calc_post = saved


def test_remap_of_a_real_read_with_the_trained_model(oracle):
    """`chunkify raw_remap` end to end on one of the reference's example reads (fixture reads.npz) with the trained pretrained.pkl
    weights: the read is remapped to the basecall stored in its fast5 file, then cut into labelled chunks.  The oracle runs the
    same chain on the posterior the device network produced."""
    import os
    need = pytest.importorskip("torch")
    from tests.gpu_util import need_gpu
    need_gpu()
    from sloika_amd import batch, chunkify_raw as cr, models, util
    batch.init_chunk_identity_worker(5, b"ACGT")
    g = np.load(os.path.join(GOLD, "reads.npz"))
    dig, off, rng, _rate = g["meta_5"]
    signal = ((g["adc_5"].astype(np.float64) + off) * (rng / dig)).astype(np.float32)
    signal = util.trim_array(signal, 200, 10)                                         # chunkify_raw.py:317-318
    ref = g["called_5"].tobytes()
    calc_post = models.from_weights_npz(os.path.join(GOLD, "pretrained_weights.npz")).compile()
    prior, slip = (25.0, 25.0), 5.0
    score, table, path, seq = cr.raw_remap(ref, signal, 1e-5, 5, prior, slip, calc_post=calc_post)
    nstep = len(path)
    assert nstep == -(-len(signal) // 5) and len(seq) == len(ref) - 4 and cr.mapping_table_is_registered(signal, table)
    # the mapping walks through (nearly) the whole reference, mostly staying or stepping by one
    assert path[0] < 0.05 * len(seq) and path[-1] > 0.95 * len(seq)
    steps = np.diff(path)
    assert ((steps >= 0) & (steps <= 2)).mean() > 0.97
    # the oracle on the same posterior
    post = calc_post(batch.normalise_chunks(signal.reshape(1, -1), "per-chunk", out_layout="network"))[:, 0, :]
    want = oracle_remap.raw_remap(ref, signal, post, 1e-5, 5, prior, slip)
    assert float(score) == pytest.approx(float(want[0]), rel=2e-6)
    assert (path == want[2]).mean() > 0.999                                          # np.log's last bit may move a tie
    if np.array_equal(path, want[2]):
        for f in ("start", "length", "seq_pos", "move"):
            assert np.array_equal(table[f], want[1][f]), f
    # labelled chunks from the device's own table: labels equal the oracle's on that table
    chunk_len, ds = 2000, 5
    chunks, labels, bad = cr.raw_chunkify(signal, table, chunk_len, 5, "per-read", ds, False)
    ml = len(signal) // chunk_len
    cols = {f: table[f].astype(np.int64) for f in ("start", "length", "seq_pos", "move")}
    trimmed = oracle_remap.trim_table(cols, len(signal), 0, ml * chunk_len)
    states = oracle_remap.states_of_reference(ref, 5)[trimmed["seq_pos"]]
    assert np.array_equal(labels, oracle_remap.chunk_labels(trimmed, states, ml, chunk_len, ds))
    assert chunks.shape == (ml, chunk_len, 1) and not bad.any()
    moved = (labels > 0).mean()
    assert 0.2 < moved < 0.7, moved                                                  # ~3600 bases over ~6600 blocks of five samples
