"""csrc/gru_bar16.hip, gru_bar16d.hip, gru_bar16q.hip -- whole Gru layer with projection AND recurrence as fp16 splits -- through the C ABI, against
the oracle (float32 C port, itself pinned to the reference's layers.py by tests/test_oracle_reference_layers.py)."""
import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu
TOL = 1e-4
SHAPES = [(96, 96), (64, 64), (32, 96), (128, 96), (64, 96), (48, 32), (16, 64)]


def _params(rs, I, n, bias=True, scale=1.0):
    iW = (rs.normal(size=(3 * n, I)) / np.sqrt(I + n)).astype(np.float32)
    sW = (scale * rs.normal(size=(2 * n, n)) / np.sqrt(2 * n)).astype(np.float32)
    sW2 = (scale * rs.normal(size=(n, n)) / np.sqrt(2 * n)).astype(np.float32)
    b = rs.normal(size=3 * n).astype(np.float32) if bias else None
    return iW, sW, sW2, b


#: the execution plans of the same arithmetic: barrier-stepped with four chunks per
#: workgroup (gru_bar16.hip) and with eight (gru_bar16d.hip, what batches beyond one workgroup per CU run: forced here through
#: bits 8-9 of `reverse` so that the small cases exercise it too) and with sixteen (gru_bar16q.hip, batches beyond eight chunks per CU)
ENTRY = "slk_gru_bar16_f32"
PLAN = 0


@pytest.fixture(autouse=True, params=["slk_gru_bar16_f32", "slk_gru_bar16_f32:8", "slk_gru_bar16_f32:16"])
def _entry(request):
    global ENTRY, PLAN
    ENTRY, _, chunks = request.param.partition(":")
    PLAN = {"": 0, "8": 2, "16": 3}[chunks]
    yield


def _call(L, x, ldx, iW, sW, sW2, b, y, ldy, T, B, I, n, reverse, lens=None, zr=None):
    return getattr(L, ENTRY)(x, ldx, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), None if b is None else b.data_ptr(),
                                 y, ldy, T, B, I, n, int(reverse) | (PLAN << 8), 1, 2, None if lens is None else lens.data_ptr(),
                                 None if zr is None else zr.data_ptr(), stream())


@pytest.mark.parametrize("I,n", SHAPES)
@pytest.mark.parametrize("T,B,reverse", [(23, 9, False), (8, 4, True), (3, 2, False), (1, 1, True), (41, 5, True)])
def test_bar16_vs_oracle(oracle, I, n, T, B, reverse):
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(I + n + T)
    iW, sW, sW2, b = _params(rs, I, n, scale=2.0)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    ref = oracle.gru(x, iW, sW, sW2, b, reverse=reverse)
    xd, iWd, sWd, sW2d, bd = dev(x), dev(iW), dev(sW), dev(sW2), dev(b)
    y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
    assert _call(L, xd.data_ptr(), I, iWd, sWd, sW2d, bd, y.data_ptr(), n, T, B, I, n, reverse) == 0
    err = np.abs(y.cpu().numpy() - ref).max()
    assert err < 2e-5, err                          # float32-grade: the exact-fp32 kernels measure ~2e-6 here
    # no bias + strided input/output rows (slices of wider tensors, as birnn produces them)
    xw = torch.zeros((T, B, I + 16), device="cuda")
    xw[:, :, 8:8 + I] = xd
    yw = torch.full((T, B, n + 8), -5.0, device="cuda")
    assert _call(L, xw.data_ptr() + 8 * 4, I + 16, iWd, sWd, sW2d, None, yw.data_ptr() + 4 * 4, n + 8, T, B, I, n, reverse) == 0
    out = yw.cpu().numpy()
    np.testing.assert_allclose(out[:, :, 4:4 + n], oracle.gru(x, iW, sW, sW2, None, reverse=reverse), atol=2e-5)
    assert (out[:, :, :4] == -5.0).all() and (out[:, :, 4 + n:] == -5.0).all()


@pytest.mark.parametrize("xmag", [1e-7, 1e-3, 1e3, 1e5, 3e7])
def test_bar16_input_magnitudes(oracle, xmag):
    """The fp16 halves of x AND of every weight row are taken AFTER a per-row power-of-two scaling: inputs and weights far
    outside fp16's range (65504) or below its normal range (6e-5) keep float32-grade accuracy.  iW is scaled inversely to x
    (up to 7e5, down to 2e-9) so that the gates are exercised, not saturated."""
    torch = need_gpu()
    from sloika_amd import _lib
    I, n, T, B = 96, 96, 17, 6
    rs = np.random.RandomState(5)
    iW, sW, sW2, b = _params(rs, I, n, scale=2.0)
    x = (rs.normal(size=(T, B, I)) * xmag).astype(np.float32)
    x[3, 2, :] *= 1e-3                                       # rows of very different size inside one 16-row MFMA tile
    x[4, 1, 7] *= 50.0                                       # an outlier inside a row
    iW = (iW / xmag).astype(np.float32)
    ref = oracle.gru(x, iW, sW, sW2, b)
    y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
    xd, iWd, sWd, sW2d, bd = dev(x), dev(iW), dev(sW), dev(sW2), dev(b)
    assert _call(_lib.lib(), xd.data_ptr(), I, iWd, sWd, sW2d, bd, y.data_ptr(), n, T, B, I, n, False) == 0
    out = y.cpu().numpy()
    assert np.isfinite(out).all()
    assert np.abs(out - ref).max() < 5e-5, np.abs(out - ref).max()


def test_bar16_recurrent_weight_rows_of_any_magnitude(oracle):
    """Rows of sW / sW2 whose magnitudes differ by many orders (one neuron's incoming weights ~1e4, another's ~1e-6)."""
    torch = need_gpu()
    from sloika_amd import _lib
    I, n, T, B = 64, 64, 25, 5
    rs = np.random.RandomState(8)
    iW, sW, sW2, b = _params(rs, I, n, scale=2.0)
    sW[3] *= 1e4; sW[n + 9] *= 3e4; sW2[17] *= 1e4           # saturating gates / candidate for those neurons
    sW[20] *= 1e-6; sW2[40] *= 1e-7; sW2[41] = 0.0
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    ref = oracle.gru(x, iW, sW, sW2, b)
    y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
    xd, iWd, sWd, sW2d, bd = dev(x), dev(iW), dev(sW), dev(sW2), dev(b)
    assert _call(_lib.lib(), xd.data_ptr(), I, iWd, sWd, sW2d, bd, y.data_ptr(), n, T, B, I, n, False) == 0
    out = y.cpu().numpy()
    assert np.isfinite(out).all()
    # neurons driven by 1e4-scale rows sit on the steep part of a saturating gate: compare those with a looser bound
    assert np.abs(out - ref).max() < TOL


def test_bar16_trained_weight_magnitudes(oracle):
    """|w| up to 6 with saturating gates, as in models/pretrained.pkl.  Few steps: recurrent weights this large make the
    map chaotic, and over dozens of steps ANY two float32 evaluations (the oracle's C port and numpy included) drift apart
    by more than the layer tolerance, which would test the dynamics, not the kernel."""
    torch = need_gpu()
    from sloika_amd import _lib
    I, n, T, B = 96, 96, 8, 7
    rs = np.random.RandomState(11)
    iW, sW, sW2, b = _params(rs, I, n, scale=3.0)
    # a few weights of the size the trained model holds (pretrained.pkl: sW2 up to 6), the bulk moderate: with EVERY weight
    # that large the recurrence is chaotic and float32 evaluations in different orders diverge from each other
    for m in (sW, sW2):
        idx = rs.randint(0, m.size, size=60)
        m.reshape(-1)[idx] = rs.choice([-6.0, 5.0, 4.5], size=60)
    iW *= 2.0
    assert np.abs(sW2).max() > 4.0
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    for reverse in (False, True):
        ref = oracle.gru(x, iW, sW, sW2, b, reverse=reverse)
        y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
        xd, iWd, sWd, sW2d, bd = dev(x), dev(iW), dev(sW), dev(sW2), dev(b)
        assert _call(_lib.lib(), xd.data_ptr(), I, iWd, sWd, sW2d, bd, y.data_ptr(), n, T, B, I, n, reverse) == 0
        assert np.abs(y.cpu().numpy() - ref).max() < TOL


@pytest.mark.parametrize("I,n", [(96, 96), (64, 64)])
def test_bar16_ragged_and_saved_gates(oracle, I, n):
    """Ragged batch (each chunk must equal the call on the chunk alone at its own length, reversed scans included) and the
    training variant that also stores the activated gates z | r."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(n)
    T = 29
    lens = [29, 1, 20, 8, 28, 9, 2]
    B = len(lens)
    iW, sW, sW2, b = _params(rs, I, n, scale=2.0)
    x = np.zeros((T, B, I), dtype=np.float32)
    for bb, tb in enumerate(lens):
        x[:tb, bb] = rs.normal(size=(tb, I))
    xd, iWd, sWd, sW2d, bd = dev(x), dev(iW), dev(sW), dev(sW2), dev(b)
    ld = dev(np.asarray(lens, dtype=np.int32))
    for reverse in (False, True):
        y = torch.full((T, B, n), np.nan, dtype=torch.float32, device="cuda")
        assert _call(L, xd.data_ptr(), I, iWd, sWd, sW2d, bd, y.data_ptr(), n, T, B, I, n, reverse, lens=ld) == 0
        out = y.cpu().numpy()
        for bb, tb in enumerate(lens):
            want = oracle.gru(x[:tb, bb:bb + 1], iW, sW, sW2, b, reverse=reverse)
            np.testing.assert_allclose(out[:tb, bb:bb + 1], want, atol=2e-5, err_msg="chunk %d" % bb)
            assert np.isnan(out[tb:, bb]).all()                     # rows past a read's end are left untouched
        # saved gates: z | r of every step, checked through the identity h_t = z h_{t-1} + (1-z) c with the oracle's h
        zr = torch.full((T, B, 2 * n), np.nan, dtype=torch.float32, device="cuda")
        y2 = torch.empty((T, B, n), dtype=torch.float32, device="cuda")
        assert _call(L, xd.data_ptr(), I, iWd, sWd, sW2d, bd, y2.data_ptr(), n, T, B, I, n, reverse, zr=zr) == 0
        h = oracle.gru(x, iW, sW, sW2, b, reverse=reverse).astype(np.float64)
        np.testing.assert_allclose(y2.cpu().numpy(), h, atol=2e-5)
        hs = h[::-1] if reverse else h
        xs = x[::-1] if reverse else x
        hprev = np.concatenate([np.zeros((1, B, n)), hs[:-1]], axis=0)
        vI = xs.astype(np.float64) @ iW.astype(np.float64).T + b
        vS = hprev @ sW.astype(np.float64).T
        zz = 1.0 / (1.0 + np.exp(-(vI[..., :n] + vS[..., :n])))
        rr = 1.0 / (1.0 + np.exp(-(vI[..., n:2 * n] + vS[..., n:])))
        want = np.concatenate([zz, rr], axis=2)
        got = zr.cpu().numpy()
        got = got[::-1] if reverse else got
        np.testing.assert_allclose(got, want, atol=2e-5)


@pytest.mark.parametrize("I,n", [(96, 96), (128, 96), (64, 64)])
def test_bar16_agrees_with_exact_kernels_under_load(I, n):
    """Race screen: every CU busy, hundreds of steps (dozens of LDS ring turnovers), repeated: any operand image or ring
    slot reused too early shows up as a large error in some chunk."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    T, B = 333, 1021
    g = torch.Generator(device="cuda").manual_seed(I * n)
    x = torch.randn((T, B, I), device="cuda", generator=g)
    iW = torch.randn((3 * n, I), device="cuda", generator=g) / np.sqrt(I + n)
    sW = 2.0 * torch.randn((2 * n, n), device="cuda", generator=g) / np.sqrt(2 * n)
    sW2 = 2.0 * torch.randn((n, n), device="cuda", generator=g) / np.sqrt(2 * n)
    b = torch.randn(3 * n, device="cuda", generator=g)
    vI = torch.empty((T, B, 3 * n), device="cuda")
    assert L.slk_gemm_bias_act_f32(x.data_ptr(), I, iW.data_ptr(), b.data_ptr(), vI.data_ptr(), 3 * n, T * B, I, 3 * n, 0,
                                   stream()) == 0
    for reverse in (0, 1):
        ref = torch.empty((T, B, n), device="cuda")
        assert L.slk_gru_recurrent_f32_ex(vI.data_ptr(), sW.data_ptr(), sW2.data_ptr(), ref.data_ptr(), n, T, B, n, reverse,
                                          1, 2, 1, stream()) == 0
        for rep in range(3):
            y = torch.full((T, B, n), float("nan"), device="cuda")
            assert _call(L, x.data_ptr(), I, iW, sW, sW2, b, y.data_ptr(), n, T, B, I, n, reverse) == 0
            err = (y - ref).abs().max().item()
            assert err < 5e-5, "reverse=%d rep=%d: %g" % (reverse, rep, err)


def test_bar16_unsupported_shapes():
    need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    z = dev(np.zeros((4, 4), dtype=np.float32))
    for I, n, act in ((7, 5, 1), (96, 128, 1), (96, 96, 3), (96, 112, 1)):
        assert getattr(L, ENTRY)(z.data_ptr(), I, z.data_ptr(), z.data_ptr(), z.data_ptr(), None, z.data_ptr(), n, 1, 1,
                                     I, n, 0, act, 2, None, None, stream()) == _lib.SLK_ERR_UNSUPPORTED


def test_layer_with_weights_outside_fp16_range(oracle):
    """Through layers.Gru (whichever kernel it picks): a weight of 1e5 times inputs of 1e-5 is an ordinary pre-activation."""
    need_gpu()
    from sloika_amd import layers
    rs = np.random.RandomState(3)
    for I, n in ((96, 96), (128, 112)):                       # fused16 kernel / projection GEMM + recurrence kernel
        iW, sW, sW2, b = _params(rs, I, n)
        x = (rs.normal(size=(11, 3, I)) * 1e-5).astype(np.float32)
        iW[5, 7] = 1e5
        iW[6] *= 1e5
        g = layers.Gru(I, n, has_bias=True)
        g.set_params({"iW": iW.reshape(3, n, I), "sW": sW.reshape(2, n, n), "sW2": sW2, "b": b.reshape(3, n)})
        y = g.compile()(x)
        assert np.isfinite(y).all()
        np.testing.assert_allclose(y, oracle.run_network(g.spec(), x), atol=TOL)


@pytest.mark.parametrize("I,n", [(96, 96), (128, 96), (16, 64)])
def test_bar16_launches_are_deterministic(I, n):
    """Race screen of a different kind: a kernel whose waves exchange data through LDS without enough ordering gives different
    bits from launch to launch.  Every launch of the same inputs -- full-size batch, ragged lengths, both directions, with the
    saved gates -- must reproduce the first one exactly."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(I + n)
    T, B = 400, 1024
    iW = torch.randn(3 * n, I, device="cuda", generator=g) / np.sqrt(I + n)
    bb = torch.randn(3 * n, device="cuda", generator=g)
    sW = 2 * torch.randn(2 * n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    sW2 = 2 * torch.randn(n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    x = torch.randn(T, B, I, device="cuda", generator=g)
    lens = torch.randint(1, T + 1, (B,), device="cuda", dtype=torch.int32, generator=g)
    for reverse in (False, True):
        for lp in (None, lens):
            first = None
            for rep in range(5):
                y = torch.full((T, B, n), float("nan"), device="cuda")
                zr = torch.full((T * B, 2 * n), float("nan"), device="cuda")
                assert _call(L, x.data_ptr(), I, iW, sW, sW2, bb, y.data_ptr(), n, T, B, I, n, reverse, lens=lp, zr=zr) == 0
                got = (torch.nan_to_num(y, nan=9.0), torch.nan_to_num(zr, nan=9.0))
                if first is None:
                    first = got
                else:
                    assert torch.equal(first[0], got[0]) and torch.equal(first[1], got[1]), (reverse, lp is not None, rep)


@pytest.mark.parametrize("I,n", [(96, 96), (64, 64), (32, 96), (48, 32)])
def test_eight_chunk_plan_is_bit_identical(I, n):
    """gru_bar16d.hip (eight chunks per workgroup) computes every (neuron, chunk) pair with the same instructions in the same order as
    gru_bar16.hip: the two plans must agree bit for bit; gru_bar16q.hip (sixteen) within float32 rounding -- states and saved gates,
    ragged and reversed, batch sizes around the multiples of eight and sixteen and batches (2048 + 3, 4096 + 4 chunks) that take the eight-
    and the sixteen-chunk plan by themselves."""
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(7 * I + n)
    iW = torch.randn(3 * n, I, device="cuda", generator=g) / np.sqrt(I + n)
    bb = torch.randn(3 * n, device="cuda", generator=g)
    sW = 2 * torch.randn(2 * n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    sW2 = 2 * torch.randn(n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    for T, B in [(1, 1), (5, 7), (9, 8), (13, 17), (2, 33), (37, 2051), (6, 4100)]:
        x = torch.randn(T, B, I, device="cuda", generator=g)
        lens = torch.randint(1, T + 1, (B,), device="cuda", dtype=torch.int32, generator=g)
        for reverse in (0, 1):
            for lp in (None, lens):
                got = []
                for plan in (1, 2, 3, 0):
                    y = torch.full((T, B, n), float("nan"), device="cuda")
                    zr = torch.full((T * B, 2 * n), float("nan"), device="cuda")
                    rc = L.slk_gru_bar16_f32(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), bb.data_ptr(), y.data_ptr(), n, T, B,
                                             I, n, reverse | (plan << 8), 1, 2, None if lp is None else lp.data_ptr(), zr.data_ptr(), stream())
                    assert rc == 0
                    got.append((torch.nan_to_num(y, nan=9.0), torch.nan_to_num(zr, nan=9.0)))
                # four and eight chunks per workgroup: the same two-MFMA products (hi and lo halves of the state in different column
                # groups) in the same order -> the same bits
                assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1]), (T, B, reverse, lp is not None)
                # sixteen chunks leave no column group to spare: the three-term products agree to float32 rounding (states and gates
                # are in [-1, 1])
                for other in got[2:]:
                    assert (got[0][0] - other[0]).abs().max().item() < 3e-6 and (got[0][1] - other[1]).abs().max().item() < 3e-6, \
                        (T, B, reverse, lp is not None)


@pytest.mark.parametrize("I,n", [(80, 80), (40, 72), (20, 20), (7, 5), (100, 64), (1, 96), (128, 90), (96, 100), (60, 130)])
def test_any_gru_shape_runs_on_the_fast_plans(oracle, I, n):
    """The reference's Gru takes any (insize, size) (layers.py:952-977; every model factory has a `size=` argument, e.g.
    models/baseline_raw_gru.py:4).  Shapes without a kernel of their own run zero-padded on the next shape that has one
    (layers.gru_pad_shape): up to 96 wide with up to 128 inputs on the one-kernel plan, up to 144 wide on the fp16-split scan -- never
    on the float32 two-kernel path ("scan") -- and compute what the oracle computes for the unpadded layer, forward and reversed."""
    need_gpu()
    from sloika_amd import activation, layers
    if _entry_is_not_default():
        pytest.skip("layer-level test: once is enough")
    rs = np.random.RandomState(100 * I + n)
    iW, sW, sW2, b = _params(rs, I, n)
    Ip, npad = layers.gru_pad_shape(I, n, activation.tanh, activation.sigmoid)
    assert layers.gru_plan(Ip, npad, activation.tanh, activation.sigmoid) in ("layer", "scan16"), (I, n, Ip, npad)
    if n <= 96 and I <= 128:
        assert layers.gru_plan(Ip, npad, activation.tanh, activation.sigmoid) == "layer"
    x = rs.normal(size=(19, 6, I)).astype(np.float32)
    g = layers.Gru(I, n, has_bias=True)
    g.set_params({"iW": iW.reshape(3, n, I), "sW": sW.reshape(2, n, n), "sW2": sW2, "b": b.reshape(3, n)})
    y = g.compile()(x)
    assert y.shape == (19, 6, n)
    np.testing.assert_allclose(y, oracle.run_network(g.spec(), x), atol=TOL)
    r = layers.Reverse(g)
    np.testing.assert_allclose(r.compile()(x), oracle.run_network(r.spec(), x), atol=TOL)


def _entry_is_not_default():
    return PLAN != 0
