"""csrc/gemm_bf16x6.hip -- y = act(x W^T + b) for long rows as six bf16 MFMA terms per product (the dL/dx products of the training
step) -- through the C ABI against float64 numpy."""
import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu

_ACTS = {0: lambda v: v, 1: np.tanh, 2: lambda v: 1.0 / (1.0 + np.exp(-v))}


def _run(L, torch, x, W, b, act, ldx=None, ldy=None):
    M, K = x.shape
    N = W.shape[0]
    ldx, ldy = ldx or K, ldy or N
    xd = torch.full((M, ldx), np.nan, dtype=torch.float32, device="cuda")
    xd[:, :K] = dev(x)
    yd = torch.full((M, ldy), np.nan, dtype=torch.float32, device="cuda")
    packed = torch.empty(L.slk_pack_bf16x3_bytes(N, K), dtype=torch.uint8, device="cuda")
    assert L.slk_pack_bf16x3_f32(dev(W).data_ptr(), N, K, packed.data_ptr(), stream()) == 0
    rc = L.slk_gemm_bias_act_bf16x6(xd.data_ptr(), ldx, packed.data_ptr(), None if b is None else dev(b).data_ptr(), yd.data_ptr(),
                                    ldy, M, K, N, act, stream())
    return rc, yd.cpu().numpy()


@pytest.mark.parametrize("M,K,N", [(1000, 288, 96), (777, 1028, 96), (64, 4, 12), (130, 256, 12), (300, 432, 144), (5, 20, 160),
                                   (63, 16, 32), (65, 48, 33), (1, 1040, 96), (4099, 336, 112), (129, 36, 200)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_bf16x6_matches_float64(M, K, N, act):
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(M + K + N)
    # gradient-like rows: magnitudes from 1e-20 to 1e+3, no common scale
    x = (rs.normal(size=(M, K)) * 10.0 ** rs.uniform(-20, 3, size=(M, 1)) * 10.0 ** rs.uniform(-2, 0, size=(M, K))).astype(np.float32)
    if act:
        x = rs.normal(size=(M, K)).astype(np.float32)
    W = (rs.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rs.normal(size=N).astype(np.float32) if act else None
    rc, y = _run(L, torch, x, W, b, act, ldx=K + 4, ldy=N + 3)
    assert rc == 0
    assert np.isnan(y[:, N:]).all()
    pre = x.astype(np.float64) @ W.T.astype(np.float64) + (0.0 if b is None else b.astype(np.float64))
    want = _ACTS[act](pre)
    bound = np.abs(x).astype(np.float64) @ np.abs(W.T).astype(np.float64) + (0.0 if b is None else np.abs(b))
    err = np.abs(y[:, :N] - want)
    assert (err <= 1e-6 * bound + 1e-37).all(), float((err / (bound + 1e-300)).max())


def test_bf16x6_is_deterministic_and_ignores_what_lies_behind_a_row():
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(3)
    M, K, N = 20000, 292, 96
    x = rs.normal(size=(M, K)).astype(np.float32)
    W = rs.normal(size=(N, K)).astype(np.float32)
    rc, a = _run(L, torch, x, W, None, 0, ldx=K + 8)     # NaNs behind every row of x: K is not a multiple of the slab here
    rc2, b = _run(L, torch, x, W, None, 0, ldx=K + 8)
    assert rc == 0 and rc2 == 0 and np.array_equal(a, b) and np.isfinite(a).all()


def test_bf16x6_refuses_what_it_does_not_cover():
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    z = torch.zeros(1 << 16, device="cuda")
    f = lambda x, ldx, K, N, act=0: L.slk_gemm_bias_act_bf16x6(x.data_ptr(), ldx, z.data_ptr(), None, z.data_ptr(), max(N, 1), 8, K, N,
                                                               act, stream())
    assert f(z, 64, 62, 96) == _lib.SLK_ERR_UNSUPPORTED
    assert f(z, 66, 64, 96) == _lib.SLK_ERR_UNSUPPORTED
    assert f(z[1:], 64, 64, 96) == _lib.SLK_ERR_UNSUPPORTED
    assert f(z, 64, 64, 96, act=5) == _lib.SLK_ERR_UNSUPPORTED
    assert f(z, 32, 64, 96) == _lib.SLK_ERR_INVALID_ARG
    assert L.slk_pack_bf16x3_bytes(96, 288) == 3 * 96 * 288 * 2
    assert L.slk_pack_bf16x3_bytes(96, 1028) == 3 * 96 * 1056 * 2
