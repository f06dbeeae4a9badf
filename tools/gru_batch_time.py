"""Per-launch time of one Gru layer (96 -> 96, T' = 800) at different batch sizes, with the shader clock sampled beside it.
    python tools/gru_batch_time.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from sloika_amd import _lib, layers
    _lib.require_gpu()
    L = _lib.lib()
    T, n = 800, 96
    g = layers.Gru(n, n, has_bias=True)
    rs = np.random.RandomState(0)
    for p in g.params():
        p.set_value((rs.normal(size=p.shape) * 0.2).astype(np.float32))
    probe_stream = torch.cuda.Stream(priority=-1)
    for B in (64, 128, 256, 512, 768, 1024, 2048):
        x = torch.tanh(torch.randn((T, B, n), device="cuda"))
        for _ in range(3):
            g._forward(x, None, False)
        torch.cuda.synchronize()
        buf = torch.zeros((8, 2), dtype=torch.int64, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(16):
            g._forward(x, None, False)
            if i % 2 == 1:
                L.slk_clock_probe(buf[i // 2].data_ptr(), 16, probe_stream.cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        v = buf.cpu().numpy().astype(np.float64)
        mhz = v[:, 0] / np.maximum(v[:, 1], 1) * 100
        print("B=%4d: %.3f ms per launch, shader clock %.0f-%.0f MHz" % (B, e0.elapsed_time(e1) / 16, mhz.min(), mhz.max()))


if __name__ == "__main__":
    main()
