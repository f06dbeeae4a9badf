"""End to end on REAL reads: the reference's example fast5 reads (fixture tests/golden/reads.npz), the reference's trained
model (models/pretrained.pkl weights, fixture pretrained_weights.npz), whole-read mode as bin/basecall_network.py raw runs
it (sloika/basecall.py:88-121).  There is no Theano to produce the reference's own calls, so the yardstick is external:
the 1D basecall ONT's production software left in the same fast5 files.  Two independent basecallers agree to ~85 % on
such reads; a wrong layer formula, gate layout, k-mer order or decoder gives ~50 % (unrelated sequences)."""
import os

import numpy as np
import pytest

from tests.gpu_util import need_gpu

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def edit_distance(a, b):
    a, b = np.frombuffer(a.encode(), dtype=np.uint8), np.frombuffer(b.encode(), dtype=np.uint8)
    idx = np.arange(len(b) + 1)
    prev = idx.copy()
    for i, ca in enumerate(a, 1):
        cur = np.minimum(prev[:-1] + (b != ca), prev[1:] + 1)
        cur = np.concatenate(([i], cur))
        prev = np.minimum.accumulate(cur - idx) + idx          # insertions: cur[j] = min(cur[j], cur[j-1] + 1)
    return int(prev[-1])


def test_edit_distance_helper():
    assert edit_distance("ACGT", "ACGT") == 0 and edit_distance("ACGT", "AGT") == 1
    assert edit_distance("AAAA", "TTTT") == 4 and edit_distance("ACGTACGT", "TACGTACG") == 2


@pytest.mark.parametrize("n", [5, 3])
def test_trained_model_on_real_reads_agrees_with_stored_basecall(n):
    need_gpu()
    from sloika_amd import basecall, bio, models
    g = np.load(os.path.join(GOLDEN, "reads.npz"))
    net = models.from_weights_npz(os.path.join(GOLDEN, "pretrained_weights.npz"))
    calc_post = net.compile()
    dig, off, rng, _rate = g["meta_%d" % n]
    signal = (g["adc_%d" % n].astype(np.float64) + off) * (rng / dig)                 # fast5.Fast5.get_read(raw=True)
    stored = g["called_%d" % n].tobytes().decode()
    kmers = bio.all_kmers(5)
    seqs = []
    for _ in range(2):
        name, score, call, nsamp = basecall.raw_read_worker(calc_post, signal, trim=(200, 10), kmer_len=5, skip=5.0,
                                                            name="read%d" % n)
        assert nsamp == len(signal) - 210 - (len(signal) - 210) % 1 or nsamp > 0
        seqs.append(bio.kmers_to_sequence([kmers[i] for i in call], always_move=True))
    assert seqs[0] == seqs[1]                                                        # deterministic
    seq = seqs[0]
    assert set(seq) <= set("ACGT") and 0.85 * len(stored) < len(seq) < 1.15 * len(stored)
    identity = 1.0 - edit_distance(seq, stored) / max(len(seq), len(stored))
    assert identity > 0.80, identity                                                 # measured: 0.851 / 0.859
    # unrelated sequence of the same composition for scale
    rs = np.random.RandomState(n)
    shuffled = "".join(rs.permutation(list(stored)))
    assert 1.0 - edit_distance(seq, shuffled) / max(len(seq), len(shuffled)) < 0.62


def test_fp16_split_projections_do_not_change_the_call():
    """The time-parallel projections run as 3-term fp16 splits (float32 accumulation); forcing plain fp32 MFMA everywhere
    (what SLOIKA_AMD_EXACT_F32=1 selects) must give the same bases on a real read -- scores agree to ~1e-6 relative."""
    need_gpu()
    from sloika_amd import basecall, bio, layers, models
    g = np.load(os.path.join(GOLDEN, "reads.npz"))
    calc_post = models.from_weights_npz(os.path.join(GOLDEN, "pretrained_weights.npz")).compile()
    dig, off, rng, _rate = g["meta_5"]
    signal = (g["adc_5"].astype(np.float64) + off) * (rng / dig)
    kmers = bio.all_kmers(5)
    saved = (layers.SPLIT_F16, layers.Softmax.split_f16)
    out = []
    try:
        for split in (True, False):
            layers.SPLIT_F16 = layers.Softmax.split_f16 = split
            _, score, call, _ = basecall.raw_read_worker(calc_post, signal, kmer_len=5, skip=5.0, name="read5")
            out.append((float(score), bio.kmers_to_sequence([kmers[i] for i in call], always_move=True)))
    finally:
        layers.SPLIT_F16, layers.Softmax.split_f16 = saved
    assert out[0][1] == out[1][1]
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-5) and out[0][0] != 0.0


def _check_against_single_reads(bc, calc_post, reads, scores, paths, lens, nsamp, skip):
    """Every read of the padded batch against (1) the same decoder on that read ALONE (a batch of one: bit for bit -- this is
    the property that makes batching exact) and (2) the reference-shaped whole-read worker basecall.raw_read_worker, which
    goes through the materialised posterior (calc_post -> decode_post, sloika/basecall.py:117-119): bit for bit when the batch
    decoder is the logits decoder too, and to float32 accumulation accuracy when it is the fused kernel of
    csrc/softmax_viterbi.hip (its log-posteriors differ from the posterior path's in the last bits)."""
    from sloika_amd import basecall, layers
    last = bc.network.layers[-1]
    kp = (last.insize + 15) // 16 * 16                               # odd widths are decoded from the zero-padded rows of a Gru twin
    fused = bc.fused_decode and type(last) is layers.Softmax and \
        last.viterbi_pack(bc.nbase, bc.kmer_len, kpad=None if kp == last.insize else kp) is not None
    for b, r in enumerate(reads):
        s1, p1, l1, n1 = bc.call_reads([r])
        assert n1[0] == nsamp[b]
        assert int(l1[0]) == int(lens[b]) and p1.cpu().numpy()[0, :int(l1[0])].tolist() == paths[b, :lens[b]].tolist(), b
        assert float(s1[0]) == float(scores[b]), b
        assert (paths[b, lens[b]:] == -1).all()
        _, score1, call1, nw = basecall.raw_read_worker(calc_post, r, trim=(0, 0), kmer_len=5, skip=skip, name="r%d" % b)
        assert nw == nsamp[b]
        if fused:
            assert float(scores[b]) == pytest.approx(float(score1), rel=2e-6, abs=1e-3), b
            same = paths[b, :lens[b]].tolist() == [int(c) for c in call1]
            assert same or abs(int(lens[b]) - len(call1)) <= max(2, len(call1) // 500), b     # a near tie may resolve the other way
        else:
            assert int(lens[b]) == len(call1) and paths[b, :lens[b]].tolist() == [int(c) for c in call1], b
            assert float(scores[b]) == float(score1), b


def test_ragged_batch_of_reads_equals_one_by_one():
    """Whole reads of different lengths in one padded batch (pipeline.Basecaller.call_reads: per-read lengths through the
    conv stride, reversed GRU scans starting at each read's own end, per-read Viterbi) must reproduce, bit for bit, what
    every read gets on its own in whole-read mode (basecall.raw_read_worker = the reference's raw_worker, batch 1)."""
    need_gpu()
    from sloika_amd import basecall, models, pipeline
    g = np.load(os.path.join(GOLDEN, "reads.npz"))
    net = models.from_weights_npz(os.path.join(GOLDEN, "pretrained_weights.npz"))
    calc_post = net.compile()
    sig = {}
    for n in (5, 3):
        dig, off, rng, _rate = g["meta_%d" % n]
        sig[n] = ((g["adc_%d" % n].astype(np.float64) + off) * (rng / dig)).astype(np.float32)
    reads = [sig[5], sig[3][1000:7777], sig[5][:9001], sig[3], sig[5][20000:20640], sig[3][30000:33003]]
    bc = pipeline.Basecaller(net, kmer_len=5, min_prob=1e-5, skip=5.0)
    scores, paths, lens, nsamp = bc.call_reads(reads)
    scores, paths, lens = scores.cpu().numpy(), paths.cpu().numpy(), lens.cpu().numpy()
    assert nsamp == [len(r) - len(r) % 100 for r in reads]          # trim_open_pore keeps whole 100-sample windows
    _check_against_single_reads(bc, calc_post, reads, scores, paths, lens, nsamp, skip=5.0)
    # the same reads in another order and batch composition give the same calls
    scores2, paths2, lens2, _ = bc.call_reads([reads[3], reads[0]])
    assert paths2.cpu().numpy()[1, :int(lens2[1])].tolist() == paths[0, :lens[0]].tolist()
    assert float(scores2[0]) == float(scores[3])


@pytest.mark.parametrize("model", ["raw_0.98_rgrgr", "baseline_raw_gru", "raw_1.00_rGr"])
def test_ragged_batch_other_architectures(model):
    """Same property through the fused GRU layer kernel (rgrgr: five 96-wide layers of alternating direction), through
    birnn stacks with FeedForward layers in between (stride 2), and through the zero-padded 110/142-wide layers."""
    need_gpu()
    from sloika_amd import basecall, models, pipeline
    net = models.randomise_zero_layers(models.build_model(model, klen=5, sd=0.5, seed=7))
    calc_post = net.compile()
    chunks = pipeline.synthetic_chunks(5, chunk_len=4000, seed=99)
    reads = [chunks[0], chunks[1][:1700], chunks[2][:3333], chunks[3][:800], np.concatenate([chunks[4], chunks[0][:1234]])]
    bc = pipeline.Basecaller(net, kmer_len=5, min_prob=1e-5, skip=0.0)
    scores, paths, lens, nsamp = bc.call_reads(reads)
    scores, paths, lens = scores.cpu().numpy(), paths.cpu().numpy(), lens.cpu().numpy()
    _check_against_single_reads(bc, calc_post, reads, scores, paths, lens, nsamp, skip=0.0)


def test_bucketed_whole_read_mode_equals_single_reads():
    """pipeline.Basecaller.call_reads_bucketed: many reads bucketed by length, ragged batches alternating over streams -- every
    read gets, bit for bit, what a batch of one gives; the buckets respect the waste bound."""
    need_gpu()
    from sloika_amd import models, pipeline
    net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=17))
    rs = np.random.RandomState(4)
    lens = rs.randint(900, 6000, size=23)
    lens[3] = lens[7] = 2500
    base = pipeline.synthetic_chunks(4, chunk_len=7000, seed=12)
    reads = [np.ascontiguousarray(base[i % 4][rs.randint(0, 900):][:n]) for i, n in enumerate(lens)]
    scores, paths, nsamp, stats = pipeline.Basecaller.call_reads_bucketed(net, reads, max_batch=6, max_waste=0.1, in_flight=2,
                                                                          kmer_len=5, skip=0.0)
    # (the streamed flow buckets by RAW length: the bound holds for those, trimming then takes up to two windows off a read -- a lot
    #  for reads this short, next to nothing for real ones)
    assert stats["reads"] == 23 and stats["batches"] >= 4 and 0.0 <= stats["padded_step_waste"] <= 0.1 + 200.0 / 900.0
    buckets = pipeline.Basecaller.length_buckets(nsamp, 6, 0.1)
    assert sorted(i for b in buckets for i in b) == list(range(23)) and max(len(b) for b in buckets) <= 6
    bc = pipeline.Basecaller(net, kmer_len=5, skip=0.0)
    for i in (0, 3, 7, 11, 22):
        s1, p1, l1, n1 = bc.call_reads([reads[i]])
        assert n1[0] == nsamp[i]
        assert p1.cpu().numpy()[0, :int(l1[0])].tolist() == paths[i].tolist(), i
        assert float(s1[0]) == float(scores[i]), i
    # that was the streamed flow (buckets by raw length, trimming on the device, no round trip to the host in front of the network);
    # the flow that trims first and buckets by trimmed length forms other batches and gives the same bits
    assert stats.get("streamed")
    s3, p3, n3, st3 = pipeline.Basecaller.call_reads_bucketed(net, reads, max_batch=6, max_waste=0.1, in_flight=2, stream_buckets=False,
                                                              kmer_len=5, skip=0.0)
    assert not st3.get("streamed") and st3["batches"] >= 4 and list(n3) == list(nsamp) and st3["padded_step_waste"] <= 0.1
    assert np.array_equal(s3, scores) and all(a.tolist() == b.tolist() for a, b in zip(p3, paths))
    # ... and with samples trimmed off both ends (util.trim_array, basecall.py:112) the two flows agree as well
    s4, p4, n4, _ = pipeline.Basecaller.call_reads_bucketed(net, reads, trim=(37, 112), max_batch=6, max_waste=0.1, kmer_len=5, skip=0.0)
    s5, p5, n5, _ = pipeline.Basecaller.call_reads_bucketed(net, reads, trim=(37, 112), max_batch=6, max_waste=0.1, kmer_len=5, skip=0.0,
                                                            stream_buckets=False)
    assert list(n4) == list(n5) and n4[0] < nsamp[0] and np.array_equal(s4, s5)
    assert all(a.tolist() == b.tolist() for a, b in zip(p4, p5))


@pytest.mark.parametrize("stream_buckets", [True, False])
@pytest.mark.parametrize("fraction", [0.0, 0.3])
def test_a_read_that_fails_is_skipped_and_the_others_are_unaffected(fraction, stream_buckets, capsys):
    """sloika/basecall.py:103-115: the reference's worker reports a read it cannot call and returns None; the pool goes on.  Here
    such a read (a NaN or an infinity among its samples, fewer samples than one open-pore window) must not poison the ragged batch it
    would have shared: it is left out, reported, and every other read gets bit for bit what it gets without it."""
    need_gpu()
    from sloika_amd import batch, models, pipeline
    net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=17))
    rs = np.random.RandomState(9)
    base = pipeline.synthetic_chunks(4, chunk_len=7000, seed=31)
    good = [np.ascontiguousarray(base[i % 4][rs.randint(0, 500):][:n]) for i, n in enumerate(rs.randint(1500, 5000, size=9))]
    nan_read = good[2].copy()
    nan_read[1234] = np.nan
    inf_read = good[5].copy()
    inf_read[7] = np.inf
    reads = good[:3] + [nan_read] + good[3:6] + [good[0][:60]] + good[6:] + [inf_read]
    bad_idx = [3, 7, len(reads) - 1]
    kw = dict(max_batch=4, max_waste=0.2, in_flight=2, kmer_len=5, skip=0.0, open_pore_fraction=fraction, stream_buckets=stream_buckets)
    scores, paths, nsamp, stats = pipeline.Basecaller.call_reads_bucketed(net, reads, **kw)
    assert bool(stats.get("streamed")) == (stream_buckets and fraction == 0.0)
    assert stats["failed"] == bad_idx and pipeline.Basecaller.failed_reads(nsamp) == bad_idx
    err = capsys.readouterr().err
    for i in bad_idx:
        assert paths[i] is None and np.isnan(scores[i]) and nsamp[i] == 0
        assert "Failure calling read %d" % i in err
    s2, p2, n2, st2 = pipeline.Basecaller.call_reads_bucketed(net, good, **kw)
    assert st2["failed"] == []
    keep = [i for i in range(len(reads)) if i not in bad_idx]
    assert [nsamp[i] for i in keep] == list(n2)
    for j, i in enumerate(keep):
        assert paths[i].tolist() == p2[j].tolist() and float(scores[i]) == float(s2[j])
    # the vectorised bounds of fraction 0 are the reference's np.percentile(., 0) bounds
    if fraction == 0.0:
        dev, off, lens = batch.upload_reads_windowed(good)
        fast = batch.open_pore_bounds_many(dev, off, lens, 0.0)
        slow = batch.open_pore_bounds_many(dev, off, lens, 1e-12)          # the general path; same threshold for all practical purposes
        assert fast == slow


def test_open_pore_trim_kernel_equals_numpy_percentile_zero():
    """slk_open_pore_trim_f32 through the C ABI against the reference's arithmetic done in numpy (sloika/batch.py:213-220 with
    max_op_fraction 0: np.percentile(spread, 0) is the minimum; then util.trim_array, basecall.py:112): spreads with ties at the minimum,
    reads whose windows are all equal, reads shorter than a window, trims that leave nothing, a read flagged as not finite."""
    torch = need_gpu()
    from sloika_amd import _lib
    from tests.gpu_util import stream
    L = _lib.lib()
    rs = np.random.RandomState(11)
    window = 100
    nwin = np.array([0, 1, 2, 5, 17, 64, 65, 400, 1150, 3, 3, 9, 30, 7], dtype=np.int32)
    first_win = np.concatenate([[0], np.cumsum(nwin)[:-1]]).astype(np.int64)
    spread = rs.gamma(2.0, size=int(nwin.sum())).astype(np.float32)
    sp = lambda r: spread[first_win[r]: first_win[r] + nwin[r]]
    sp(3)[:] = 1.5                                                       # all windows equal: no window livelier than the minimum
    sp(4)[[0, 3, 16]] = sp(4).min()                                      # ties at the minimum, at both ends
    sp(9)[:] = [2.0, 1.0, 2.0]
    sp(10)[:] = [1.0, 2.0, 1.0]                                          # one lively window in the middle
    first_sample = (first_win * window + 7 * np.arange(len(nwin))).astype(np.int64)      # reads need not be packed tightly
    flags_in = np.zeros(len(nwin), dtype=np.int32)
    flags_in[12] = 1                                                     # the caller found a sample that is not finite
    for trim in ((0, 0), (37, 112), (250, 250)):
        d = lambda a: torch.from_numpy(a).cuda()
        start = torch.full((len(nwin),), -1, dtype=torch.int64, device="cuda")
        ln = torch.full((len(nwin),), -1, dtype=torch.int32, device="cuda")
        flags = d(flags_in.copy())
        sd, fw, nw, fs = d(spread), d(first_win), d(nwin), d(first_sample)
        rc = L.slk_open_pore_trim_f32(sd.data_ptr(), fw.data_ptr(), nw.data_ptr(), fs.data_ptr(), len(nwin), window, trim[0], trim[1],
                                      start.data_ptr(), ln.data_ptr(), flags.data_ptr(), stream())
        assert rc == 0
        start, ln, flags = start.cpu().numpy(), ln.cpu().numpy(), flags.cpu().numpy()
        for r in range(len(nwin)):
            want_flag, lo, hi = int(flags_in[r]), 0, 0
            s = sp(r)
            lively = np.flatnonzero(s > np.percentile(s, 0)) if len(s) else np.zeros(0, dtype=np.int64)
            if len(lively) == 0:
                want_flag |= 2                                           # the reference's function raises on such a read
            else:
                lo, hi = int(lively[0]) * window + trim[0], (int(lively[-1]) + 1) * window - trim[1]
                if hi - lo < 1:
                    want_flag |= 4
            assert flags[r] == want_flag, (r, trim, flags[r], want_flag)
            if want_flag:
                assert ln[r] == 0 and start[r] == first_sample[r]
            else:
                assert ln[r] == hi - lo and start[r] == first_sample[r] + lo, (r, trim)
    z = torch.zeros(4, dtype=torch.int64, device="cuda")
    assert L.slk_open_pore_trim_f32(None, z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 100, 0, 0, z.data_ptr(), z.data_ptr(), z.data_ptr(),
                                    stream()) == _lib.SLK_ERR_INVALID_ARG
    assert L.slk_open_pore_trim_f32(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 100, -1, 0, z.data_ptr(), z.data_ptr(),
                                    z.data_ptr(), stream()) == _lib.SLK_ERR_INVALID_ARG
