"""GPU parity: chunk normalisation (bit exact), conv1d, window, activations -- through the C ABI."""
import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu


def test_normalise_bit_exact_vs_golden_and_oracle(oracle, golden_signal):
    need_gpu()
    from sloika_amd import batch
    g = golden_signal
    chunks = g["chunks_none"]
    out = batch.normalise_chunks(chunks, 'per-chunk')
    assert np.array_equal(out, g["chunks_per_chunk"])                       # reference raw_chunkify output
    out2, med, mad = batch.normalise_chunks(chunks, 'per-read', return_stats=True)
    assert np.array_equal(out2, g["chunks_per_read"])
    net = batch.normalise_chunks(chunks, 'per-chunk', out_layout='network')
    assert net.shape == (4000, 5, 1)
    assert np.array_equal(net[:, :, 0].T, g["chunks_per_chunk"])
    net2 = batch.normalise_chunks(chunks, 'per-read', out_layout='network')
    assert np.array_equal(net2[:, :, 0].T, g["chunks_per_read"])
    assert np.array_equal(batch.normalise_chunks(chunks, 'none'), chunks)


@pytest.mark.parametrize("n,clen", [(1, 1), (3, 2), (7, 3), (4, 63), (5, 64), (3, 65), (2, 1000), (9, 4096), (2, 5000)])
def test_normalise_ragged_sizes_vs_oracle(oracle, n, clen):
    need_gpu()
    from sloika_amd import batch
    rs = np.random.RandomState(n * 1000 + clen)
    x = (rs.normal(size=(n, clen)) * 10 + 80).astype(np.float32)
    if clen > 8:
        x[0, :clen // 2] = x[0, 0]                     # many duplicates -> ties in the sort
    with np.errstate(all="ignore"):
        ref, rmed, rmad = oracle.med_mad_normalise(x, return_stats=True)
    out, med, mad = batch.normalise_chunks(x, 'per-chunk', return_stats=True)
    assert np.array_equal(med, rmed) and np.array_equal(mad, rmad)
    assert np.array_equal(out, ref, equal_nan=True)


@pytest.mark.parametrize("clen", [1024, 1025, 2047, 2048, 2049, 3000, 3999, 4000, 4001, 4095, 4096])
@pytest.mark.parametrize("kind", ["normal", "rounded", "two-valued", "negative", "constant", "hundredths", "wide"])
def test_normalise_by_selection_vs_oracle(oracle, clen, kind):
    """Chunks of 1024..4096 samples take med_mad_select_kernel (order statistics by bitwise selection on register-resident
    keys): odd and even lengths, both register footprints, heavy duplication (the upper median neighbour is then the same
    key), signals of both signs, both ways the lower 16 bits are finished (on the collected keys / on all of them) -- median, MAD
    and the normalised samples bit for bit."""
    need_gpu()
    from sloika_amd import batch
    rs = np.random.RandomState(clen)
    x = (rs.normal(size=(37, clen)) * 12 + 90).astype(np.float32)
    if kind == "rounded":
        x = np.round(x)
    elif kind == "two-valued":
        x = np.where(rs.uniform(size=x.shape) < 0.5, 1.0, 2.0).astype(np.float32)
        x[:, 0] = 5.0
    elif kind == "negative":
        x = (x - 90.0).astype(np.float32)
    elif kind == "constant":                                            # every key shares every bit: the selection's collected keys overflow
        x[:] = 93.25
        x[1, ::7] = 93.5
    elif kind == "hundredths":                                          # a few hundred keys share the answer's upper 16 bits
        x = (np.round(x * 100) / 100).astype(np.float32)
    elif kind == "wide":                                                # magnitudes over twelve octaves: few keys share upper bits
        x = (x * np.exp2(rs.randint(-6, 7, size=x.shape))).astype(np.float32)
    with np.errstate(all="ignore"):
        ref, rmed, rmad = oracle.med_mad_normalise(x, return_stats=True)
    out, med, mad = batch.normalise_chunks(x, 'per-chunk', return_stats=True)
    assert np.array_equal(med, rmed) and np.array_equal(mad, rmad)
    assert np.array_equal(out, ref, equal_nan=True)


def test_normalise_rejects_bad_shapes():
    need_gpu()
    from sloika_amd import batch
    with pytest.raises(ValueError):
        batch.normalise_chunks(np.zeros(10, dtype=np.float32))


@pytest.mark.parametrize("T,B,Cin,Cout,w,s,mode,act", [
    (100, 3, 1, 8, 11, 5, 'same', "elu"),
    (4000, 4, 1, 96, 11, 5, 'same', "elu"),            # rgrgr front end
    (4000, 3, 1, 64, 11, 2, 'same', "tanh"),           # baseline_raw_gru front end
    (100, 20, 12, 32, 11, 5, 'same', "tanh"),          # test_layers.py Convolution(12,32,11,5) on [100,20,12]
    (30, 2, 3, 4, 4, 1, 'same', "linear"),
    (30, 2, 3, 4, 4, 1, 'same_left', "linear"),
    (30, 1, 2, 3, 5, 3, 'valid', "relu"),
    (17, 2, 2, 3, 5, 1, 'full', "sigmoid"),
    (64, 2, 128, 200, 11, 3, 'half', "tanh"),          # filter too large for LDS -> global path
    (4000, 2, 1, 128, 11, 5, (5, 5), "elu"),           # pretrained.pkl front end, explicit padding (vec4 kernel, 32 quads)
    (4000, 2, 1, 32, 11, 2, 'same', "tanh"),           # bigger_raw_gru front end (vec4 kernel, 8 quads -> 8 positions/step)
    (333, 3, 1, 4, 16, 16, 'valid', "relu"),           # one quad, widest window and stride the Cin=1 kernels take
    (57, 2, 1, 256, 3, 1, 'full', "sigmoid"),          # 64 quads = one position per step
    (41, 2, 1, 20, 7, 3, 'same_left', "softplus"),     # vec4 kernel with a run-time activation
    (41, 2, 1, 10, 7, 3, 'same', "tanh"),              # Cout not a multiple of 4 -> one-feature-per-lane kernel
    (5, 1, 1, 12, 11, 5, 'same', "linear"),            # shorter than the window
])
def test_conv1d_vs_oracle(oracle, T, B, Cin, Cout, w, s, mode, act):
    need_gpu()
    from sloika_amd import layers, activation
    rs = np.random.RandomState(T + Cin + Cout)
    x = rs.normal(size=(T, B, Cin)).astype(np.float32)
    layer = layers.Convolution(Cin, Cout, w, s, has_bias=True, fun=getattr(activation, act), padding_mode=mode)
    layer.set_params({"W": (rs.normal(size=(Cout, Cin, w)) * 0.3).astype(np.float32),
                      "b": rs.normal(size=Cout).astype(np.float32)})
    y = layer.compile()(x)
    ref = oracle.run_network(layer.spec(), x)
    assert y.shape == ref.shape
    np.testing.assert_allclose(y, ref, atol=2e-5)


def test_conv1d_chunk_major_input_equals_network_layout(oracle):
    torch = need_gpu()
    from sloika_amd import layers, activation
    rs = np.random.RandomState(5)
    chunks = rs.normal(size=(6, 500)).astype(np.float32)
    layer = layers.Convolution(1, 16, 11, 5, has_bias=True, fun=activation.elu)
    layer.set_params({"W": rs.normal(size=(16, 1, 11)).astype(np.float32) * 0.3, "b": rs.normal(size=16).astype(np.float32)})
    cd = dev(chunks)
    y1 = layer.run_strided(cd.data_ptr(), 500, 6, 1, 500, cd.device)
    y2 = layer.run(dev(np.ascontiguousarray(chunks.T)[:, :, None]))
    assert torch.equal(y1, y2)


def test_window_exact(oracle):
    need_gpu()
    from sloika_amd import layers
    x = np.random.RandomState(1).normal(size=(25, 2, 3)).astype(np.float32)
    for w in (1, 3, 5):
        y = layers.Window(3, w).compile()(x)
        assert np.array_equal(y, oracle.window(x, w))
    with pytest.raises(AssertionError):
        layers.Window(3, 2)                            # layers.py:328-329


def test_all_activations_vs_oracle(oracle):
    need_gpu()
    from sloika_amd import activation
    x = np.linspace(-6, 6, 193).astype(np.float32)
    for name in oracle.ACTIVATIONS:
        y = getattr(activation, name)(x)
        np.testing.assert_allclose(y, oracle.activation(name, x), rtol=2e-6, atol=2e-6, err_msg=name)


def test_trim_open_pore_matches_reference(golden_signal):
    """batch.trim_open_pore (sloika/batch.py:194-220): slice bounds equal the reference's."""
    need_gpu()
    from sloika_amd import batch
    g = golden_signal
    sig = g["signal"]
    for frac in (0.0, 0.3):
        lo, hi = g["trim_open_pore_%g" % frac]
        out = batch.trim_open_pore(sig, frac)
        assert out.base is sig or out.base is sig.base or np.shares_memory(out, sig)
        assert len(out) == hi - lo and np.array_equal(out, sig[lo:hi])
        lo, hi = g["trim_open_pore_std_%g" % frac]                    # var_method='std' (batch.py:210-211)
        out = batch.trim_open_pore(sig, frac, var_method='std')
        assert len(out) == hi - lo and np.array_equal(out, sig[lo:hi])
    with pytest.raises(AssertionError):
        batch.trim_open_pore(sig, 0.3, var_method='variance')


def test_window_std_kernel_vs_numpy():
    torch = need_gpu()
    from sloika_amd import _lib
    rs = np.random.RandomState(3)
    for nwin, win in ((1, 1), (7, 100), (513, 100), (40, 333)):
        x = (rs.normal(size=(nwin, win)) * rs.uniform(0.1, 30, size=(nwin, 1)) + 90).astype(np.float32)
        out = torch.empty(nwin, dtype=torch.float32, device="cuda")
        assert _lib.lib().slk_window_std_f32(dev(x).data_ptr(), nwin, win, out.data_ptr(), stream()) == 0
        np.testing.assert_allclose(out.cpu().numpy(), x.astype(np.float64).std(axis=1), rtol=2e-7)


@pytest.mark.parametrize("n", [32769, 70001, 114400])
def test_whole_read_normalisation_radix_select(oracle, n):
    """Reads longer than the LDS sort (e.g. data/reads/read1: 114400 samples, test_fast5.py:103) take the exact
    radix-selection path; bit-identical to the numpy semantics restated by the oracle."""
    need_gpu()
    from sloika_amd import batch
    rs = np.random.RandomState(n)
    sig = (rs.normal(size=n) * 12 + 90).astype(np.float32)
    sig[: n // 3] = np.round(sig[: n // 3])            # many duplicates
    sig[5] = -3.5                                       # negative keys too
    ref, rmed, rmad = oracle.med_mad_normalise(sig[None, :], return_stats=True)
    out, med, mad = batch.normalise_chunks(sig[None, :], 'per-chunk', return_stats=True)
    assert med[0] == rmed[0] and mad[0] == rmad[0]
    assert np.array_equal(out, ref)


def test_raw_read_worker_whole_read(oracle):
    """Whole-read path of sloika/basecall.py:110-121 (batch 1, per-read normalisation) against the oracle."""
    torch = need_gpu()
    from sloika_amd import models, basecall, util, pipeline
    net = models.build_model("raw_0.98_rgrgr", seed=6)
    sig = pipeline.synthetic_chunks(1, chunk_len=6210, seed=12)[0]
    res = basecall.raw_read_worker(net.compile(), sig, trim=(200, 10), open_pore_fraction=0.0, kmer_len=5, skip=0.0,
                                   name="r1")
    from sloika_amd import batch
    trimmed = util.trim_array(batch.trim_open_pore(sig, 0.0), 200, 10)      # bounds pinned by the golden test above
    assert res is not None and res[0] == "r1" and res[3] == len(trimmed) and 5900 <= len(trimmed) <= 6000
    x = oracle.med_mad_normalise(trimmed[None, :])
    post = oracle.run_network(net.spec(), np.ascontiguousarray(x.T)[:, :, None])
    assert post.shape == ((len(trimmed) + 4) // 5, 1, 1025)
    # decode of the oracle posterior gives (almost always) the same call; scores agree to float32 accuracy
    o_score, o_call = oracle.viterbi(oracle.prepare_post(post, 1e-5), 5, skip_pen=0.0)
    assert abs(float(res[1]) - float(o_score)) < 1e-2 * max(1.0, abs(float(o_score)) * 1e-3)
    same = sum(a == b for a, b in zip(res[2], o_call))
    assert len(res[2]) == len(o_call) and same >= 0.99 * len(o_call)
    assert basecall.raw_read_worker(net.compile(), sig[:205], trim=(200, 10)) is None


def test_elu_keeps_relative_accuracy_near_zero():
    """activation.elu = expm1 on the negative side (sloika/activation.py:52-57).  exp(x) - 1 through the hardware exponential
    is only accurate to an ulp of 1.0; tiny pre-activations (and the gradients that pass through y + 1 in training) need the
    relative accuracy of expm1."""
    need_gpu()
    from sloika_amd import activation
    x = -np.logspace(-9, 1, 400).astype(np.float32)
    y = activation.elu(x)
    ref = np.expm1(x.astype(np.float64))
    assert np.abs(y - ref).max() < 2.5e-7
    assert (np.abs(y - ref) / np.abs(ref)).max() < 1e-5
    small = np.abs(x) < 1e-2
    assert (np.abs(y[small] - ref[small]) / np.abs(ref[small])).max() < 1e-6
    assert np.array_equal(activation.elu(np.array([0.0, 2.5, 1e-8], dtype=np.float32)), np.array([0.0, 2.5, 1e-8], dtype=np.float32))
