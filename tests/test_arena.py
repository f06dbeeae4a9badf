"""sloika_amd.device.Arena: the buffers a Basecaller(borrow=True) keeps across calls (the reference's th.function(borrow=True),
sloika/layers.py:34-36).  The bookkeeping is device-independent, so it is checked here on host tensors: the k-th request of a pass gets
the k-th buffer, a repeated pass allocates nothing, results alternate between two sets, a bigger request grows its buffer only."""
import torch

from sloika_amd import device as D

CPU = torch.device("cpu")


def test_requests_of_a_pass_get_their_buffers_back_and_results_alternate():
    a = D.Arena(generations=2)
    ptrs = []
    for rnd in range(4):
        with a:
            x = D.scratch((3, 4), torch.float32, CPU)
            w = D.scratch(100, torch.uint8, CPU)
            r = D.scratch((2, 5), torch.int32, CPU, result=True)
            assert x.shape == (3, 4) and x.dtype == torch.float32 and w.numel() == 100 and r.shape == (2, 5) and r.dtype == torch.int32
            assert x.is_contiguous() and r.is_contiguous()
            ptrs.append((x.data_ptr(), w.data_ptr(), r.data_ptr()))
    assert ptrs[0][:2] == ptrs[1][:2] == ptrs[2][:2] == ptrs[3][:2]              # layer buffers: the same memory every pass
    assert ptrs[0][2] != ptrs[1][2] and ptrs[0][2] == ptrs[2][2] and ptrs[1][2] == ptrs[3][2]    # results: two sets, in turn
    assert a.grown == 4                                                         # x, w and two result sets: nothing after pass 2


def test_a_bigger_request_grows_only_its_own_buffer_and_outside_a_pass_nothing_is_kept():
    a = D.Arena()
    with a:
        x = D.scratch((8,), torch.float32, CPU)
        y = D.scratch((8,), torch.float32, CPU)
    grown = a.grown
    with a:
        x2 = D.scratch((4,), torch.float32, CPU)               # smaller: a view of the same buffer
        y2 = D.scratch((64,), torch.float32, CPU)              # bigger: a new buffer for this position only
    assert x2.data_ptr() == x.data_ptr() and y2.data_ptr() != y.data_ptr() and a.grown == grown + 1
    # no pass active: plain fresh tensors
    p, q = D.scratch((8,), torch.float32, CPU), D.scratch((8,), torch.float32, CPU)
    assert p.data_ptr() != q.data_ptr() and a.grown == grown + 1
    # passes nest (an inner arena does not leak into the outer one's sequence)
    b = D.Arena()
    with a:
        x3 = D.scratch((4,), torch.float32, CPU)
        with b:
            z = D.scratch((4,), torch.float32, CPU)
        y3 = D.scratch((64,), torch.float32, CPU)
    assert x3.data_ptr() == x.data_ptr() and y3.data_ptr() == y2.data_ptr() and z.data_ptr() not in (x3.data_ptr(), y3.data_ptr())
