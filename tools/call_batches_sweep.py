#!/usr/bin/env python3
"""pipeline.Basecaller.call_batches: samples/s over a few seconds for a model, a batch size and several numbers of batches in flight
(deterministic plans only).    python tools/call_batches_sweep.py [model] [batch] [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch  # noqa: E402
from sloika_amd import models, pipeline  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "raw_0.98_rgrgr"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
L = 4000
net = models.randomise_zero_layers(models.build_model(model, klen=5, sd=0.5, seed=11))
dev = [torch.from_numpy(pipeline.synthetic_chunks(B, chunk_len=L, seed=1 + i)).cuda() for i in range(2)]
for nfl in (1, 2, 4, 8):
    slots = pipeline.Basecaller.batch_slots(net, nfl, kmer_len=5, skip=0.0)

    def feed(n=None, t0=None):
        k = 0
        while (k < n) if n is not None else (time.perf_counter() - t0 < secs):
            k += 1
            yield dev[k & 1]
    sum(1 for _ in pipeline.Basecaller.call_batches(net, feed(n=3 * nfl), slots=slots, copy=False))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = sum(1 for _ in pipeline.Basecaller.call_batches(net, feed(t0=t0), slots=slots, copy=False))
    torch.cuda.synchronize()
    d = time.perf_counter() - t0
    print("%s B=%d in flight %d: %.0f M samples/s (%d batches, %.2f ms per batch)" % (model, B, nfl, B * L * n / d / 1e6, n, d / n * 1e3), flush=True)
    del slots
    torch.cuda.empty_cache()
