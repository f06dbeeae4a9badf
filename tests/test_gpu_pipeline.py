"""GPU parity of whole networks and of the end-to-end hot path, stage by stage against the oracle.

Stage 1: posteriors within 1e-4 absolute of the oracle (north_star tolerance on fp32 layer outputs).
Stage 2: the decoder run on the GPU's own posteriors is bit-identical to the oracle's decoder on the same numbers.
"""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN
from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _logits_reference(oracle, net, x):
    """(hidden state after the last recurrent layer, logits) of the oracle: the float32 C port for the layer stack, the
    final projection in float64."""
    spec = net.spec()
    body = {"type": "serial", "sublayers": spec["sublayers"][:-1]}
    hid = oracle.run_network(body, x)
    last = spec["sublayers"][-1]
    logits = hid.astype(np.float64) @ last["W"].astype(np.float64).T + last["b"].astype(np.float64)
    return hid, logits


def _check_end_to_end(oracle, net, chunks, klen=5, skip=0.0, tol=TOL):
    torch = need_gpu()
    from sloika_amd import _lib, pipeline, decode
    bc = pipeline.Basecaller(net, kmer_len=klen, skip=skip, fused_decode=False)
    cd = dev(chunks)
    post = bc.posteriors(cd)
    x = oracle.med_mad_normalise(chunks)
    xin = np.ascontiguousarray(x.T)[:, :, None]
    ref = oracle.run_network(net.spec(), xin)
    assert post.shape == ref.shape
    got = post.cpu().numpy()
    err = np.abs(got - ref).max()
    assert err < tol, "posterior max abs err %g" % err
    # 1025-way posteriors average 1e-3: the absolute bound alone would let a 10 % error through.  Relative to each row's
    # largest posterior, and on the quantities BEFORE the softmax squashes them: hidden state and logits.
    assert (np.abs(got - ref) / ref.max(axis=2, keepdims=True)).max() < 2e-4
    hid_ref, logits_ref = _logits_reference(oracle, net, xin)
    hid = bc._hidden(cd, len(net.layers) - 1)
    assert np.abs(hid.cpu().numpy() - hid_ref).max() < 2e-5
    logits, stats, ld = net.layers[-1].logits_and_stats(hid)
    lg = logits.view(hid.shape[0], hid.shape[1], ld)[:, :, :net.size].cpu().numpy()
    assert np.abs(lg - logits_ref).max() <= 1e-4 * np.abs(logits_ref).max()
    scores, paths, lens = bc.call_chunks(cd)                       # logits path: posterior never materialised
    s2, p2, l2 = decode.viterbi_batch(post, klen, skip_pen=skip, min_prob=1e-5)   # posterior path
    assert torch.equal(paths, p2) and torch.equal(lens, l2) and torch.equal(scores, s2)
    lp = torch.empty_like(post)
    _lib.check(_lib.lib().slk_log_post_f32(post.data_ptr(), lp.data_ptr(), post.numel(), _lib.POST_RAW, 1e-5, stream()))
    o_scores, o_paths, o_lens = oracle.viterbi_batch(lp.cpu().numpy(), klen, skip_pen=skip)
    assert np.array_equal(lens.cpu().numpy(), o_lens)
    assert np.array_equal(paths.cpu().numpy(), o_paths)
    assert np.array_equal(scores.cpu().numpy(), o_scores)
    _check_fused_decode(oracle, net, cd, lp, klen, skip)
    return err


def _check_fused_decode(oracle, net, cd, lp_two_kernel, klen=5, skip=0.0, pick=None):
    """The default path: decoding straight from the Softmax layer's input (csrc/softmax_viterbi.hip).  Its log-posteriors
    (dumped) agree with those of the two-kernel path, and its paths / scores are the oracle decoder's on ITS log-posteriors,
    bit for bit.  `pick`: chunk indices to check (all when None)."""
    torch = need_gpu()
    from sloika_amd import pipeline
    bcf = pipeline.Basecaller(net, kmer_len=klen, skip=skip)
    last = net.layers[-1]
    # a width between the kernel's own (raw_1.00_rGr: 110; 80; 72) is decoded from rows with zero columns up to the next of them
    kp = next((k for k in pipeline.Basecaller.FUSED_WIDTHS if k >= last.insize), None)
    if kp is None or last.viterbi_pack(4, klen, kpad=None if kp == last.insize else kp) is None:
        return                                                       # shape outside the fused kernel: nothing more to check
    T, B = lp_two_kernel.shape[0], cd.shape[0]
    dump = torch.empty((T, B, last.size), dtype=torch.float32, device="cuda")
    scores, paths, lens = bcf.call_chunks(cd, lp_dump=dump)
    s0, p0, l0 = bcf.call_chunks(cd)                                 # the instantiation without the dump
    assert torch.equal(scores, s0) and torch.equal(paths, p0) and torch.equal(lens, l0)
    sel = np.arange(B) if pick is None else pick
    d = dump[:, torch.from_numpy(sel).cuda(), :].contiguous()
    assert (d - lp_two_kernel).abs().max().item() < 2e-5
    o_scores, o_paths, o_lens = oracle.viterbi_batch(d.cpu().numpy(), klen, skip_pen=skip)
    assert np.array_equal(lens.cpu().numpy()[sel], o_lens)
    assert np.array_equal(paths.cpu().numpy()[sel], o_paths)
    assert np.array_equal(scores.cpu().numpy()[sel], o_scores)


@pytest.mark.parametrize("name,nchunk,chunk_len", [("raw_0.98_rgrgr", 6, 1000), ("baseline_raw_gru", 5, 600),
                                                   ("bigger_raw_gru", 3, 400), ("raw_1.00_rGr", 2, 300)])
def test_raw_models_end_to_end(oracle, name, nchunk, chunk_len):
    from sloika_amd import models, pipeline
    net = models.randomise_zero_layers(models.build_model(name, klen=5, sd=0.5, seed=21))
    chunks = pipeline.synthetic_chunks(nchunk, chunk_len=chunk_len, seed=3)
    _check_end_to_end(oracle, net, chunks)


@pytest.mark.parametrize("width,feed", [(80, False), (72, True), (100, False), (120, True)])
def test_fused_decode_takes_any_width_up_to_128(oracle, width, feed):
    """A Softmax layer whose input is not 64 / 96 / 112 / 128 wide is still decoded by the fused kernel: from the zero-padded rows
    of the Gru twin that made them (no copy), or -- behind a FeedForward layer -- from one padded copy of the hidden state."""
    torch = need_gpu()
    from sloika_amd import layers, activation, pipeline
    rs = np.random.RandomState(width)
    init = lambda shape: (rs.normal(size=shape) * 0.3).astype(np.float32)
    stack = [layers.Convolution(1, 48, 11, 5, init=init, has_bias=True, fun=activation.tanh),
             layers.Gru(48, width, init=init, has_bias=True, fun=activation.tanh)]
    if feed:
        stack.append(layers.FeedForward(width, width, init=init, has_bias=True, fun=activation.tanh))
    stack.append(layers.Softmax(width, 1025, init=init, has_bias=True))
    net = layers.Serial(stack)
    chunks = pipeline.synthetic_chunks(5, chunk_len=600, seed=width)
    bcf = pipeline.Basecaller(net, kmer_len=5)
    cd = dev(chunks)
    hid = bcf._hidden(cd, len(net.layers) - 1)
    packed = bcf._fused_pack(net.layers[-1], hid)
    assert packed is not None and packed[1].shape[2] in pipeline.Basecaller.FUSED_WIDTHS       # the fused kernel is what runs
    assert (packed[1].data_ptr() == hid.data_ptr()) == (not feed)                                # ... without a copy behind a Gru
    _check_end_to_end(oracle, net, chunks)


def test_pretrained_rgr_end_to_end(oracle):
    """Trained weights of models/pretrained.pkl (|w| up to 6): the hard case for fp32 parity."""
    from sloika_amd import models, pipeline
    net = models.from_weights_npz(os.path.join(GOLDEN, "pretrained_weights.npz"))
    chunks = pipeline.synthetic_chunks(4, chunk_len=1500, seed=5)
    _check_end_to_end(oracle, net, chunks, skip=0.0)


@pytest.mark.parametrize("name", ["tiny_gru", "baseline_gru", "baseline_lstm"])
def test_event_models_posteriors(oracle, name):
    """Events route (config 0 plumbing): Window front end, birnn, [T,1,4] features."""
    need_gpu()
    from sloika_amd import models
    net = models.randomise_zero_layers(models.build_model(name, klen=5, sd=0.5, seed=5))
    x = np.random.RandomState(2).normal(size=(120, 2, 4)).astype(np.float32)
    y = net.compile()(x)
    ref = oracle.run_network(net.spec(), x)
    assert y.shape == (120, 2, 1025)
    np.testing.assert_allclose(y, ref, atol=TOL)
    assert np.allclose(y.sum(axis=2), 1.0, atol=1e-5)


def test_compile_contract_numpy_and_tensor(oracle):
    """Layer.compile() -> callable(ndarray [T,B,F]) -> ndarray [T',B,size]  (layers.py:34-36)."""
    torch = need_gpu()
    from sloika_amd import models
    net = models.build_model("raw_0.98_rgrgr", seed=2)
    f = net.compile()
    x = np.random.RandomState(0).normal(size=(250, 3, 1)).astype(np.float32)
    y = f(x)
    assert isinstance(y, np.ndarray) and y.dtype == np.float32 and y.shape == (50, 3, 1025)
    yt = f(torch.from_numpy(x).cuda())
    assert isinstance(yt, torch.Tensor) and yt.is_cuda
    assert np.array_equal(yt.cpu().numpy(), y)


def test_raw_chunk_worker_and_seqprinter(oracle, golden_bio, golden_decode):
    """Worker tuple (name, score, call, n) of sloika/basecall.py:88-121 + SeqPrinter text (basecall.py:157-163)."""
    import io
    need_gpu()
    from sloika_amd import models, pipeline, basecall
    net = models.build_model("raw_0.98_rgrgr", seed=4)
    chunks = pipeline.synthetic_chunks(3, chunk_len=500, seed=9)
    res = basecall.raw_chunk_worker(net.compile(), chunks, 5, min_prob=1e-5, skip=0.0)
    assert len(res) == 3 and all(len(r) == 4 and r[3] == 500 for r in res)
    sp = basecall.SeqPrinter(5, datatype="samples", transducer=True, alphabet="ACGT")
    sp.fh = io.StringIO()
    g = golden_bio["seqprinter"]
    nb = sp.write(g["read_name"], g["score"], [int(v) for v in golden_decode[g["path_key"]]], g["nev"])
    assert sp.fh.getvalue() == g["text"] and nb == g["nbases"]


@pytest.mark.parametrize("name,B", [("raw_0.98_rgrgr", 1024), ("baseline_raw_gru", 256), ("raw_0.98_rgrgr", 2048),
                                    ("bigger_raw_gru", 1024), ("raw_0.98_rgrgr", 4096)])
def test_full_size_batch_sampled_chunks_vs_oracle(oracle, name, B):
    """BASELINE.json configs[2] / configs[1] at FULL size (4000-sample chunks, batch 1024 / 256): eight chunks picked at
    random out of the batch are compared with the oracle run on those chunks alone -- posteriors relative to each row's
    maximum, and the decoded paths against the oracle's decoder on the device's own log-posteriors.  Batch 2048 is two
    batches handed over as one call (bench.py's `as_one_batch`): the recurrent layers run eight chunks per workgroup and the
    decoder one wave per chunk; batch 4096 is four batches as one call (sixteen chunks per workgroup).  bigger_raw_gru at batch 1024 (configs[3] per GPU) runs the two directions of each birnn side
    by side, eight chunks per workgroup each."""
    torch = need_gpu()
    from sloika_amd import _lib, models, pipeline
    net = models.randomise_zero_layers(models.build_model(name, klen=5, sd=0.5, seed=13))
    bc = pipeline.Basecaller(net, fused_decode=False)
    chunks = pipeline.synthetic_chunks(B, chunk_len=4000, seed=77)
    cd = dev(chunks)
    post = bc.posteriors(cd)
    scores, paths, lens = bc.call_chunks(cd)
    pick = np.sort(np.random.RandomState(B).choice(B, size=8, replace=False))
    sub = post[:, torch.from_numpy(pick).cuda(), :].contiguous()
    got = sub.cpu().numpy()
    x = oracle.med_mad_normalise(chunks[pick])
    ref = oracle.run_network(net.spec(), np.ascontiguousarray(x.T)[:, :, None])
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-5
    assert (np.abs(got - ref) / ref.max(axis=2, keepdims=True)).max() < 3e-4
    lp = torch.empty_like(sub)
    _lib.check(_lib.lib().slk_log_post_f32(sub.data_ptr(), lp.data_ptr(), sub.numel(), _lib.POST_RAW, 1e-5, stream()))
    o_scores, o_paths, o_lens = oracle.viterbi_batch(lp.cpu().numpy(), 5, skip_pen=0.0)
    assert np.array_equal(lens.cpu().numpy()[pick], o_lens)
    assert np.array_equal(paths.cpu().numpy()[pick], o_paths)
    assert np.array_equal(scores.cpu().numpy()[pick], o_scores)
    del post
    _check_fused_decode(oracle, net, cd, lp, pick=pick)


def test_full_size_batch_properties():
    """BASELINE config shape (4000-sample chunks, rgrgr) at B=64: results do not depend on batch composition
    (each chunk decoded alone == decoded inside the batch), which is what makes read sharding exact."""
    torch = need_gpu()
    from sloika_amd import models, pipeline
    net = models.build_model("raw_0.98_rgrgr", seed=8)
    bc = pipeline.Basecaller(net)
    chunks = pipeline.synthetic_chunks(64, chunk_len=4000, seed=1)
    s_all, p_all, l_all = bc.call_chunks(dev(chunks))
    assert p_all.shape == (64, 800)
    for lo, hi in ((0, 4), (13, 14), (60, 64)):
        s, p, l = bc.call_chunks(dev(chunks[lo:hi]))
        assert torch.equal(p, p_all[lo:hi]) and torch.equal(l, l_all[lo:hi]) and torch.equal(s, s_all[lo:hi])


def test_double_batch_equals_two_batches():
    """One set of bits per chunk, whatever the plan.  Two batches of 1024 chunks as ONE call take other kernels than each on its own
    (csrc/gru_bar16d.hip: eight chunks per workgroup, the same two-MFMA products in the same order) and give the same bits: paths,
    lengths and scores.  Four batches as one call would run sixteen chunks per workgroup (csrc/gru_bar16q.hip, no column group to
    spare: three-term products, states equal to float32 rounding only); the default Basecaller (deterministic=True) keeps them on
    the eight-chunk plan, so they too give the SAME bits.  Basecaller(deterministic=False) lets the faster plan in: scores agree to
    1e-5 relative and the calls except where two paths score within rounding of each other."""
    torch = need_gpu()
    from sloika_amd import models, pipeline
    net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=21))
    bc = pipeline.Basecaller(net)
    chunks = dev(pipeline.synthetic_chunks(4096, chunk_len=2000, seed=5))
    s4, p4, l4 = bc.call_chunks(chunks)                      # four batches as one call: still eight chunks per workgroup (two rounds)
    s2, p2, l2 = bc.call_chunks(chunks[:2048])               # two: eight
    for lo in (0, 1024):
        s1, p1, l1 = bc.call_chunks(chunks[lo:lo + 1024])    # one: four
        assert torch.equal(p1, p2[lo:lo + 1024]) and torch.equal(l1, l2[lo:lo + 1024]) and torch.equal(s1, s2[lo:lo + 1024])
    assert torch.equal(p2, p4[:2048]) and torch.equal(l2, l4[:2048]) and torch.equal(s2, s4[:2048])
    # ... and four batches in flight on streams of their own (the hint that used to select the sixteen-chunk plan)
    s1f, p1f, l1f = pipeline.Basecaller(net, in_flight=4).call_chunks(chunks[:1024])
    assert torch.equal(p1f, p4[:1024]) and torch.equal(l1f, l4[:1024]) and torch.equal(s1f, s4[:1024])
    # the faster plan, on request: agreement to rounding
    fast = pipeline.Basecaller(net, deterministic=False)
    sq, pq, lq = fast.call_chunks(chunks)                    # sixteen chunks per workgroup
    assert torch.allclose(s2, sq[:2048], rtol=1e-5, atol=0.0)
    same = ((p2 == pq[:2048]).all(dim=1) & (l2 == lq[:2048])).float().mean().item()
    assert same > 0.98, same                                 # whole chunks called identically


def test_in_flight_hint_changes_the_plan_not_the_result():
    """Basecaller(in_flight=2) runs the Gru layers of a 1024-chunk batch eight chunks per workgroup (half the chip per batch):
    same bits as the default."""
    torch = need_gpu()
    from sloika_amd import models, pipeline
    net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=22))
    chunks = dev(pipeline.synthetic_chunks(1024, chunk_len=1500, seed=6))
    s1, p1, l1 = pipeline.Basecaller(net).call_chunks(chunks)
    s2, p2, l2 = pipeline.Basecaller(net, in_flight=2).call_chunks(chunks)
    assert torch.equal(p1, p2) and torch.equal(l1, l2) and torch.equal(s1, s2)
    # a birnn model at the batch north_star quotes, four in flight: the two directions of each birnn run side by side on the
    # eight-chunk plan (4 batches x 2 directions x 32 workgroups = the device's CUs)
    net = models.randomise_zero_layers(models.build_model("baseline_raw_gru", klen=5, sd=0.5, seed=23))
    chunks = dev(pipeline.synthetic_chunks(256, chunk_len=1200, seed=7))
    s1, p1, l1 = pipeline.Basecaller(net).call_chunks(chunks)
    s4, p4, l4 = pipeline.Basecaller(net, in_flight=4).call_chunks(chunks)
    assert torch.equal(p1, p4) and torch.equal(l1, l4) and torch.equal(s1, s4)


def test_first_calls_on_two_streams_without_warm_up():
    """Two Basecallers sharing one network, each on a stream of its own, first calls issued back to back with nothing warmed
    up: the device caches derived from the weights (fp16 splits, packed softmax weights, padded twins) are built on the
    stream of the first caller, and the second stream must wait for them (layers._derived_cache records an event)."""
    torch = need_gpu()
    from sloika_amd import models, pipeline
    for name, B in (("raw_0.98_rgrgr", 64), ("raw_1.00_rGr", 8), ("baseline_raw_gru", 32)):
        chunks = dev(pipeline.synthetic_chunks(B, chunk_len=2000, seed=31))
        net_ref = models.randomise_zero_layers(models.build_model(name, klen=5, sd=0.5, seed=41))
        ref = pipeline.Basecaller(net_ref).call_chunks(chunks)
        torch.cuda.synchronize()
        net = models.randomise_zero_layers(models.build_model(name, klen=5, sd=0.5, seed=41))     # same weights, cold caches
        bcs = [pipeline.Basecaller(net, in_flight=2) for _ in range(2)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = []
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                outs.append(bcs[k].call_chunks(chunks))
        torch.cuda.synchronize()
        for got in outs:
            assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]) and torch.equal(got[0], ref[0]), name


def test_event_model_through_the_basecaller(oracle):
    """Event-feature models take the [T, B, features] tensor itself (basecall.py:73-75); raw chunks are refused instead of being
    read with the wrong feature count."""
    torch = need_gpu()
    from sloika_amd import decode, models, pipeline
    net = models.randomise_zero_layers(models.build_model("baseline_lstm", klen=5, sd=0.5, seed=5))
    x = np.random.RandomState(4).normal(size=(90, 3, 4)).astype(np.float32)
    bc = pipeline.Basecaller(net, kmer_len=5, nbase=4, min_prob=1e-5, skip=0.0)
    scores, paths, lens = bc.call_chunks(torch.from_numpy(x).cuda())
    post = net.compile()(x)
    s2, p2, l2 = decode.viterbi_batch(post, 5, skip_pen=0.0, nbase=4, min_prob=1e-5)
    assert torch.equal(lens.cpu(), l2.cpu())
    agree = (paths.cpu() == p2.cpu()).float().mean().item()
    assert agree > 0.98, agree                       # the fused decoder's log-posteriors differ from the posterior path's in the last bits
    with pytest.raises(ValueError):
        bc.call_chunks(pipeline.synthetic_chunks(2, chunk_len=200, seed=1))
    with pytest.raises(ValueError):
        bc.call_chunks(torch.zeros((10, 2, 3), device="cuda"))


def test_borrow_mode_allocates_nothing_after_the_first_call_and_changes_no_bit():
    """Basecaller(borrow=True): the reference compiles its networks with borrow=True (layers.py:34-36) -- what a call returns may be
    overwritten by a later one.  Here every buffer of a call comes out of the Basecaller's arena (device.Arena): the second call of the
    same shape allocates NOTHING (neither the arena nor torch's caching allocator grows), the results are those of the default
    Basecaller bit for bit, and what a call returned stays intact through the NEXT call (two result sets) -- the bench copies paths to
    the host while the next call already runs."""
    torch = need_gpu()
    from sloika_amd import models, pipeline
    for name, B, L in (("raw_0.98_rgrgr", 256, 2000), ("baseline_raw_gru", 64, 1000), ("pretrained", 128, 1500)):
        net = models.randomise_zero_layers(models.build_model(name, klen=5, sd=0.5, seed=31))
        a = dev(pipeline.synthetic_chunks(B, chunk_len=L, seed=7))
        b = dev(pipeline.synthetic_chunks(B, chunk_len=L, seed=8))
        ref_a = [t.clone() for t in pipeline.Basecaller(net).call_chunks(a)]
        ref_b = [t.clone() for t in pipeline.Basecaller(net).call_chunks(b)]
        bc = pipeline.Basecaller(net, borrow=True)
        ra = bc.call_chunks(a)
        rb = bc.call_chunks(b)                       # second result set: ra is still intact
        torch.cuda.synchronize()
        for x, y in zip(ra, ref_a):
            assert torch.equal(x, y)
        for x, y in zip(rb, ref_b):
            assert torch.equal(x, y)
        grown, reserved, nalloc = bc._arena.grown, torch.cuda.memory_reserved(), torch.cuda.memory_stats()["allocation.all.allocated"]
        for _ in range(3):
            rc = bc.call_chunks(a)
        torch.cuda.synchronize()
        assert bc._arena.grown == grown and torch.cuda.memory_reserved() == reserved
        assert torch.cuda.memory_stats()["allocation.all.allocated"] == nalloc, name       # not one allocator request in three calls
        for x, y in zip(rc, ref_a):
            assert torch.equal(x, y)
        assert rc[1].data_ptr() == rb[1].data_ptr() or rc[1].data_ptr() == ra[1].data_ptr()   # ... the two sets alternate
        del bc, ra, rb, rc


@pytest.mark.parametrize("name,B,L,nbatch,nfl", [("baseline_raw_gru", 64, 1000, 7, 3), ("raw_0.98_rgrgr", 128, 1500, 5, 8),
                                                 ("raw_0.98_rgrgr", 96, 1000, 9, 2)])
def test_call_batches_keeps_order_and_bits(name, B, L, nbatch, nfl):
    """pipeline.Basecaller.call_batches: a stream of batches with `nfl` of them in flight on a fixed set of streams (what the reference
    does with a pool of workers, bin/basecall_network.py:100-104).  One result per batch, in input order, on the host, bit for bit what
    call_chunks gives for that batch alone -- with more slots than batches, with fewer, from device tensors and from host arrays."""
    torch = need_gpu()
    from sloika_amd import models, pipeline
    net = models.randomise_zero_layers(models.build_model(name, klen=5, sd=0.5, seed=41))
    host_batches = [pipeline.synthetic_chunks(B if i % 3 else B - 8, chunk_len=L, seed=100 + i) for i in range(nbatch)]
    ref = []
    bc = pipeline.Basecaller(net)
    for hb in host_batches:
        s, p, l = bc.call_chunks(dev(hb))
        ref.append((s.cpu().numpy(), p.cpu().numpy(), l.cpu().numpy()))
    for source in ("device", "host"):
        feed = (dev(hb) if source == "device" else hb for hb in host_batches)
        got = list(pipeline.Basecaller.call_batches(net, feed, in_flight=nfl))
        assert len(got) == nbatch
        for (s, p, l), (rs, rp, rl) in zip(got, ref):
            assert np.array_equal(l, rl) and np.array_equal(p, rp) and np.array_equal(s, rs), source
    assert list(pipeline.Basecaller.call_batches(net, iter(()), in_flight=nfl)) == []
    # a server keeps its slots over several streams of batches: the second stream allocates nothing and gives the same bits
    slots = pipeline.Basecaller.batch_slots(net, nfl)
    for rnd in range(5):                     # (a slot is warm once it has filled both of its result sets)
        if rnd == 4:
            torch.cuda.synchronize()
            before = torch.cuda.memory_stats()["allocation.all.allocated"]
        got = list(pipeline.Basecaller.call_batches(net, (dev(hb) for hb in host_batches[:3] * 2), slots=slots))
        for j, (s, p, l) in enumerate(got):
            assert np.array_equal(p, ref[j % 3][1]) and np.array_equal(s, ref[j % 3][0])
    torch.cuda.synchronize()
    # (the uploads of the host batches are the caller's: 6 tensors; the calls themselves add none)
    assert torch.cuda.memory_stats()["allocation.all.allocated"] - before <= 6
