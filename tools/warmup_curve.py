#!/usr/bin/env python3
"""How long a fresh process takes to reach its steady state: the device time of every one of the first N steps of bench.py's Runner
(an event behind each step on the launch stream, no host synchronisation in between), and the shader clock sampled every few steps.

    python tools/warmup_curve.py [N] [--sleep-ms MS]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 200
t_start = time.perf_counter()
run = bench.Runner(torch, "raw_0.98_rgrgr", 1024, 4000, 1)
torch.cuda.synchronize()
print("runner built in %.2f s" % (time.perf_counter() - t_start))
probe = bench.ClockProbe(torch, nmax=64)
spin_ms = float(sys.argv[sys.argv.index("--spin-ms") + 1]) if "--spin-ms" in sys.argv else 0.0
if spin_ms > 0:
    a = torch.randn(4096, 4096, device="cuda")
    ts = time.perf_counter()
    k = 0
    while (time.perf_counter() - ts) * 1e3 < spin_ms:
        b = a @ a
        k += 1
        if k % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("spun %d matmuls in %.1f ms" % (k, (time.perf_counter() - ts) * 1e3))
from sloika_amd import profiler  # noqa: E402
rec = profiler.start() if "--stages" in sys.argv else None
noprobe = "--no-probe" in sys.argv
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
ev[0].record()
t0 = time.perf_counter()
mode = "nocopy" if "--no-copy" in sys.argv else ("samestream" if "--same-stream" in sys.argv else "bench")
for i in range(n):
    if mode == "bench":
        if "--throttle" in sys.argv and run.copied_before[0] is not None:
            run.copied_before[0].synchronize()          # the host has consumed the paths of step i - 2 before it issues step i
        if "--throttle1" in sys.argv and run.copied[0] is not None:
            run.copied[0].synchronize()
        run.step(i)
    else:
        scores, paths, lens = run.bcs[0].call_chunks(run.dev[i % run.nbuf])
        if mode == "samestream":
            run.out_host[0][0][:, : paths.shape[1]].copy_(paths, non_blocking=True)
    ev[i + 1].record()
    host.append(time.perf_counter() - t0)
    if i % 8 == 0 and not noprobe:
        probe.sample()
torch.cuda.synchronize()
if rec is not None:
    profiler.stop()
    per = len(rec.records) // n
    for k in range(min(n, 30)):
        print("step %2d: %s" % (k, " ".join("%s:%.0f" % (nm[:9], e0.elapsed_time(e1) * 1e3) for nm, e0, e1, *_ in rec.records[k * per:(k + 1) * per])))
wall = time.perf_counter() - t0
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("wall %.1f ms for %d steps; host issue of step k done at: %s" % (wall * 1e3, n, " ".join("%.1f" % (h * 1e3) for h in host[:12])))
for lo in range(0, n, 10):
    print("steps %3d-%3d: %s" % (lo, lo + 9, " ".join("%.3f" % v for v in ms[lo:lo + 10])))
print("clock", probe.result())
