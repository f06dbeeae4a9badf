"""Hardware check of the operand layout / CBSZ-ABID broadcast of v_mfma_f32_4x4x1_16b_f32 that the recurrent
GRU kernel (csrc/recurrent.hip) is built on."""
import ctypes as C

import numpy as np
import pytest

from tests.gpu_util import need_gpu, dev, stream

pytestmark = pytest.mark.gpu


def _probe(a, b, cbsz, abid):
    torch = need_gpu()
    from sloika_amd import _lib
    L = _lib.lib()
    fn = L.slk_selftest_mfma4_f32
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    d = torch.zeros(256, dtype=torch.float32, device="cuda")
    ad, bd = dev(a.astype(np.float32)), dev(b.astype(np.float32))
    assert fn(ad.data_ptr(), bd.data_ptr(), d.data_ptr(), cbsz, abid, stream()) == 0
    torch.cuda.synchronize()
    return d.cpu().numpy().reshape(4, 64)          # [vgpr i][lane]


def _expected(a, b, cbsz, abid):
    """D[i][lane = 4*blk + j] = A[src_blk(blk)*4 + i] * B[4*blk + j]; with CBSZ, blocks are grouped in
    2^cbsz and every block of a group reads A from block (group_base + abid)."""
    out = np.zeros((4, 64), dtype=np.float32)
    g = 1 << cbsz
    for blk in range(16):
        src = (blk // g) * g + abid if cbsz else blk
        for i in range(4):
            for j in range(4):
                out[i, 4 * blk + j] = a[4 * src + i] * b[4 * blk + j]
    return out


@pytest.mark.parametrize("cbsz,abid", [(0, 0), (4, 0), (4, 3), (4, 15), (3, 0), (3, 5), (2, 1), (2, 3)])
def test_mfma_4x4x1_layout(cbsz, abid):
    rs = np.random.RandomState(cbsz * 16 + abid)
    a = rs.randint(1, 50, size=64).astype(np.float32)
    b = (rs.randint(1, 50, size=64) * 64 + np.arange(64)).astype(np.float32)      # asymmetric, lane-identifying
    got = _probe(a, b, cbsz, abid)
    exp = _expected(a, b, cbsz, abid)
    if not np.array_equal(got, exp):
        # print the observed mapping to make a layout surprise diagnosable from the log
        print("cbsz", cbsz, "abid", abid)
        for i in range(4):
            print("vgpr", i, "lanes 0..15 got", got[i, :16], "exp", exp[i, :16])
    assert np.array_equal(got, exp)
