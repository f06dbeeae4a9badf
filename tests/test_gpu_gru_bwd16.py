"""csrc/gru_bwd16.hip -- the Gru reverse scan on the barrier-stepped fp16-split plan -- against the float32 kernels of
csrc/train.hip / gru_backward_mfma.hip (slk_gru_backward_f32, itself pinned to the float64 training oracle by
tests/test_gpu_train.py), through the C ABI.  Gradients have no natural range: the inputs are scaled from 1e-9 to 1e+3."""
import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu


def _inputs(rs, T, B, n, scale, wscale=2.0):
    """A consistent forward pass (h_t = z h + (1 - z) c) so that the scan's candidate recovery sees what training gives it."""
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    z = sig(rs.normal(size=(T, B, n)) * 2.0)
    r = sig(rs.normal(size=(T, B, n)) * 2.0)
    c = np.tanh(rs.normal(size=(T, B, n)) * 1.5)
    z[rs.uniform(size=z.shape) < 0.02] = 1.0                          # saturated update gates: the candidate is not recoverable there
    h = np.zeros((T + 1, B, n))
    for t in range(T):
        h[t + 1] = z[t] * h[t] + (1.0 - z[t]) * c[t]
    dy = rs.normal(size=(T, B, n)) * scale * 10.0 ** rs.uniform(-3, 0, size=(T, B, 1))
    dy[rs.uniform(size=(T, B, n)) < 0.3] = 0.0
    dy[:, rs.uniform(size=B) < 0.2] *= 1e-6                            # chunks whose gradients are far below the others'
    sW = wscale * rs.normal(size=(2 * n, n)) / np.sqrt(2 * n)
    sW2 = wscale * rs.normal(size=(n, n)) / np.sqrt(2 * n)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return (f(dy), f(np.concatenate([z, r], axis=2)), f(h[1:]), f(h[:-1]), f(sW), f(sW2))


def _run(L, entry, dy, zr, hout, hprev, sW, sW2, T, B, n, reverse):
    import torch
    da = torch.full((T * B, 3 * n), float("nan"), device="cuda")
    rh = torch.full((T * B, n), float("nan"), device="cuda")
    rc = getattr(L, entry)(dy.data_ptr(), n, hprev.data_ptr(), n, zr.data_ptr(), hout.data_ptr(), n, sW.data_ptr(), sW2.data_ptr(),
                           da.data_ptr(), rh.data_ptr(), T, B, n, int(reverse), 1, 2, stream())
    return rc, da, rh


def _flip(arrs, reverse):
    """The scan walks time backwards from the LAST scan step; for a reversed layer the scan order is the time order reversed."""
    return [np.ascontiguousarray(a[::-1]) for a in arrs] if reverse else arrs


@pytest.mark.parametrize("n", [16, 32, 48, 64, 96, 112, 128])
@pytest.mark.parametrize("T,B,reverse,scale", [(23, 9, False, 1.0), (8, 4, True, 1e-9), (3, 2, False, 1e3), (1, 1, True, 1.0),
                                               (61, 5, True, 1e-4), (200, 33, False, 1e-2)])
def test_gru_bwd16_vs_fp32_kernel(n, T, B, reverse, scale):
    need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(n + T)
    dy, zr, hout, hprev, sW, sW2 = _inputs(rs, T, B, n, scale)
    dy, zr, hout, hprev = _flip([dy, zr, hout, hprev], reverse)
    args = [dev(a.reshape(T * B, -1)) for a in (dy, zr, hout, hprev)] + [dev(sW), dev(sW2)]
    rc0, want, wantr = _run(L, "slk_gru_backward_f32", *args, T, B, n, reverse)
    rc1, got, gotr = _run(L, "slk_gru_backward16_f32", *args, T, B, n, reverse)
    assert rc0 == 0 and rc1 == 0
    w, g = want.cpu().numpy().reshape(T, B, 3 * n), got.cpu().numpy().reshape(T, B, 3 * n)
    assert np.isfinite(g).all()
    # per chunk: relative to that chunk's largest gradient (what the per-chunk scaling of the kernel preserves)
    top = np.maximum(np.abs(w).max(axis=(0, 2), keepdims=True), 1e-35)
    assert (np.abs(g - w) <= 5e-5 * top).all(), float((np.abs(g - w) / top).max())
    assert np.abs(gotr.cpu().numpy() - wantr.cpu().numpy()).max() <= 1e-6


@pytest.mark.parametrize("n", [64, 96, 128])
def test_gru_bwd16_large_weights_and_repeats_bit_for_bit(n):
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    T, B = 120, 517
    rs = np.random.RandomState(3)
    dy, zr, hout, hprev, sW, sW2 = _inputs(rs, T, B, n, 1e-3, wscale=4.0)
    args = [dev(a.reshape(T * B, -1)) for a in (dy, zr, hout, hprev)] + [dev(sW), dev(sW2)]
    _, want, wantr = _run(L, "slk_gru_backward_f32", *args, T, B, n, False)
    first = None
    for rep in range(3):
        rc, got, gotr = _run(L, "slk_gru_backward16_f32", *args, T, B, n, False)
        assert rc == 0
        if first is None:
            first = (got, gotr)
        else:
            assert torch.equal(first[0], got) and torch.equal(first[1], gotr)
    w, g = want.cpu().numpy().reshape(T, B, 3 * n), first[0].cpu().numpy().reshape(T, B, 3 * n)
    assert np.isfinite(g).all()
    top = np.maximum(np.abs(w).max(axis=(0, 2), keepdims=True), 1e-35)
    assert (np.abs(g - w) <= 1e-4 * top).all(), float((np.abs(g - w) / top).max())


def test_gru_bwd16_unsupported_shapes_are_refused():
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    z = torch.zeros(4096, device="cuda")
    for n, act, gate in [(144, 1, 2), (24, 1, 2), (64, 2, 2), (64, 1, 1)]:
        assert L.slk_gru_backward16_f32(z.data_ptr(), n, z.data_ptr(), n, z.data_ptr(), z.data_ptr(), n, z.data_ptr(), z.data_ptr(),
                                        z.data_ptr(), z.data_ptr(), 1, 1, n, 0, act, gate, stream()) == _lib.SLK_ERR_UNSUPPORTED
    assert L.slk_gru_backward16_f32(None, 64, z.data_ptr(), 64, z.data_ptr(), z.data_ptr(), 64, z.data_ptr(), z.data_ptr(),
                                    z.data_ptr(), z.data_ptr(), 1, 1, 64, 0, 1, 2, stream()) == _lib.SLK_ERR_INVALID_ARG


@pytest.mark.parametrize("n,insize", [(96, 96), (96, 32), (64, 64), (48, 16), (32, 64), (96, 80), (16, 16)])
@pytest.mark.parametrize("T,B,reverse,scale,dact", [(23, 9, False, 1.0, None), (61, 5, True, 1e-4, "elu"), (1, 1, True, 1.0, "tanh"),
                                                    (200, 33, False, 1e-2, None), (8, 4, True, 1e-9, "relu")])
def test_gru_bwd16_with_dx_inside(n, insize, T, B, reverse, scale, dact):
    """slk_gru_backward16_dx_f32: the layer's dL/dx = da . iW formed in the reverse pass itself (updates.py:67's gradient through the
    projection of layers.py:1011).  da and rh are those of the plain pass BIT FOR BIT; dx agrees with the float64 product of that da
    with iW to float32 rounding of a chunk's largest entry -- times fun'(.) of the layer below's output when one is handed over."""
    torch = need_gpu()
    from sloika_amd import _lib, activation
    L = _lib.lib()
    rs = np.random.RandomState(n + insize + T)
    dy, zr, hout, hprev, sW, sW2 = _inputs(rs, T, B, n, scale)
    dy, zr, hout, hprev = _flip([dy, zr, hout, hprev], reverse)
    args = [dev(a.reshape(T * B, -1)) for a in (dy, zr, hout, hprev)] + [dev(sW), dev(sW2)]
    iW = (2.0 * rs.normal(size=(3 * n, insize)) / np.sqrt(n + insize)).astype(np.float32)
    yb = np.tanh(rs.normal(size=(T * B, insize))).astype(np.float32)
    rc0, want, wantr = _run(L, "slk_gru_backward16_f32", *args, T, B, n, reverse)
    da = torch.full((T * B, 3 * n), float("nan"), device="cuda")
    rh = torch.full((T * B, n), float("nan"), device="cuda")
    ldx = insize + 3                                                    # rows of dx need not be dense
    dx = torch.full((T * B, ldx), float("nan"), device="cuda")
    iWd, ybd = dev(iW), dev(yb)
    act_id = activation.act_id(getattr(activation, dact)) if dact else 0
    rc = L.slk_gru_backward16_dx_f32(args[0].data_ptr(), n, args[3].data_ptr(), n, args[1].data_ptr(), args[2].data_ptr(), n,
                                     args[4].data_ptr(), args[5].data_ptr(), iWd.data_ptr(), da.data_ptr(), rh.data_ptr(), dx.data_ptr(),
                                     ldx, T, B, n, insize, int(reverse), 1, 2, ybd.data_ptr() if dact else None, insize, act_id, stream())
    kernel_width = 64 if n <= 64 else 96
    if insize > kernel_width:
        assert rc == _lib.SLK_ERR_UNSUPPORTED                           # (more inputs than a wave per 16 units can form)
        return
    assert rc0 == 0 and rc == 0
    assert torch.equal(da, want) and torch.equal(rh, wantr)
    got = dx.cpu().numpy()
    assert np.isnan(got[:, insize:]).all()                              # nothing written past a row's insize floats
    ref = want.cpu().numpy().astype(np.float64) @ iW.astype(np.float64)
    if dact:
        y64 = yb.astype(np.float64)
        ref *= {"tanh": 1.0 - y64 * y64, "elu": np.where(y64 > 0, 1.0, y64 + 1.0), "relu": (y64 > 0).astype(np.float64)}[dact]
    g, w = got[:, :insize].reshape(T, B, insize), ref.reshape(T, B, insize)
    assert np.isfinite(g).all()
    top = np.maximum(np.abs(w).max(axis=(0, 2), keepdims=True), 1e-35)
    assert (np.abs(g - w) <= 2e-6 * top).all(), float((np.abs(g - w) / top).max())
