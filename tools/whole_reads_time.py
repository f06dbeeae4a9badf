#!/usr/bin/env python3
"""Where the whole-read mode's time from host arrays goes: every phase of Basecaller.prepare_read_batches timed by itself (host clock,
synchronised), then the resident run, three times over.     python tools/whole_reads_time.py [nreads]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from sloika_amd import batch, models, pipeline  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=11))
reads = bench.synthetic_reads(n)
if "--after-in-flight" in sys.argv:
    # what bench.py's in-flight legs leave behind: four Basecallers with arenas of their own, steps on four streams, a 4096-chunk call
    run = bench.Runner(torch, "raw_0.98_rgrgr", 1024, 4000, 4)
    for nact in (1, 2, 4):
        run.set_in_flight(nact)
        for i in range(8 * nact):
            run.step(i, nact)
    torch.cuda.synchronize()
    if "--big" in sys.argv:
        big = torch.cat([run.dev[i % 2] for i in range(4)], dim=0)
        run.set_in_flight(1)
        run.bcs[0].call_chunks(big)
        torch.cuda.synchronize()
        del big
    if "--release" in sys.argv:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    net = run.net
kw = dict(kmer_len=5, skip=0.0)
lanes = pipeline.Basecaller.read_lanes(net, 8, **kw)
pipeline.Basecaller.call_reads_bucketed(net, reads, max_batch=256, max_waste=0.08, lanes=lanes, **kw)
torch.cuda.synchronize()
total = sum(len(r) for r in reads)
print("%d reads, %.1f M samples, %d host cores" % (n, total / 1e6, os.cpu_count()))


def tick(label, t0):
    torch.cuda.synchronize()
    t = time.perf_counter()
    print("   %-28s %7.1f ms" % (label, (t - t0) * 1e3))
    return t


for rep in range(3):
    print("pass %d" % rep)
    t0 = t = time.perf_counter()
    dev, off, lens = batch.upload_reads_windowed(reads)
    t = tick("upload_reads_windowed", t)
    bad = batch.reads_nonfinite(dev, off, lens)
    t = tick("reads_nonfinite", t)
    bounds = batch.open_pore_bounds_many(dev, off, lens, 0.0)
    t = tick("open_pore_bounds_many", t)
    del dev
    batches, nsamp = pipeline.Basecaller.prepare_read_batches(net, reads, max_batch=256, max_waste=0.08, **kw)
    t = tick("prepare_read_batches (all)", t)
    scores, paths = pipeline.Basecaller.run_read_batches(net, batches, len(nsamp), lanes=lanes, **kw)
    t = tick("run_read_batches", t)
    del batches
    t1 = time.perf_counter()
    pipeline.Basecaller.call_reads_bucketed(net, reads, max_batch=256, max_waste=0.08, lanes=lanes, **kw)
    torch.cuda.synchronize()
    d = time.perf_counter() - t1
    print("   call_reads_bucketed          %7.1f ms = %.0f M samples/s" % (d * 1e3, sum(nsamp) / d / 1e6))
