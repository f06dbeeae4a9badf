import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch, bench
from sloika_amd import models, pipeline
net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=11))
reads = bench.synthetic_reads(4096)
kw = dict(kmer_len=5, skip=0.0)
lanes = pipeline.Basecaller.read_lanes(net, 8, **kw)
tot = sum(len(r) for r in reads)
for rep in range(6):
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    pipeline.Basecaller.call_reads_bucketed(net, reads, max_batch=256, max_waste=0.08, lanes=lanes, **kw)
    torch.cuda.synchronize()
    d = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats()
    print("call %d: %.0f ms = %.0f M/s; device mallocs %d, frees %d, retries %d, reserved %.1f GB, allocated peak %.1f GB" % (
        rep, d * 1e3, tot / d / 1e6, ms1["num_device_alloc"] - ms0.get("num_device_alloc", 0), ms1["num_device_free"] - ms0.get("num_device_free", 0),
        ms1["num_alloc_retries"] - ms0.get("num_alloc_retries", 0), ms1["reserved_bytes.all.current"] / 1e9, ms1["allocated_bytes.all.peak"] / 1e9))
