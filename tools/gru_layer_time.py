#!/usr/bin/env python3
"""Per-launch time of one Gru layer at T' = 800:  python tools/gru_layer_time.py [insize:size ...] [--batch 1024]
Prints the time of the whole layer call (projection + scan where they are separate kernels) per launch."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="*", default=["112:144", "128:112", "144:112", "96:96"])
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=800)
    ap.add_argument("--reps", type=int, default=8)
    a = ap.parse_args()
    import torch
    from sloika_amd import _lib, layers, profiler
    _lib.require_gpu()
    rs = np.random.RandomState(0)
    for shape in a.shapes:
        i, n = (int(v) for v in shape.split(":"))
        g = layers.Gru(i, n, has_bias=True)
        for p in g.params():
            p.set_value((rs.normal(size=p.shape) * 0.2).astype(np.float32))
        x = torch.tanh(torch.randn((a.steps, a.batch, i), device="cuda"))
        for rev in (False, True):
            for _ in range(2):
                g._forward(x, None, rev)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                g._forward(x, None, rev)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            print("Gru %3d -> %3d  B=%d T=%d %s: %.3f ms per call (%.0f ns per step)" % (
                i, n, a.batch, a.steps, "reverse" if rev else "forward", ms, ms * 1e6 / a.steps))


if __name__ == "__main__":
    main()
