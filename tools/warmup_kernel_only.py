#!/usr/bin/env python3
"""The first launches of ONE kernel in a fresh process (the 96-wide Gru layer at B = 1024, T' = 800; nothing else on the device, one
stream, an event between launches): does the device itself need time to reach its steady state?   [--spin-ms MS] [--sleep-ms MS]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sloika_amd import models  # noqa: E402

n = 200
net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=11))
gru = net.layers[2]
x = torch.randn(800, 1024, 96, device="cuda")
y = torch.empty_like(x)
if "--spin-ms" in sys.argv:
    ms_ = float(sys.argv[sys.argv.index("--spin-ms") + 1])
    a = torch.randn(4096, 4096, device="cuda")
    ts = time.perf_counter()
    while (time.perf_counter() - ts) * 1e3 < ms_:
        b = a @ a
        torch.cuda.synchronize()
if "--sleep-ms" in sys.argv:
    torch.cuda.synchronize()
    time.sleep(float(sys.argv[sys.argv.index("--sleep-ms") + 1]) * 1e-3)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    gru._forward(x, y, False)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for lo in range(0, n, 20):
    print("launch %3d-: %s" % (lo, " ".join("%.0f" % (v * 1e3) for v in ms[lo:lo + 20])))

# second part: does a pause re-arm the transient?  100 launches, synchronise, sleep X ms, 40 launches
for pause in (0.0, 1.0, 5.0, 20.0, 100.0):
    for i in range(100):
        gru._forward(x, y, False)
    torch.cuda.synchronize()
    if pause:
        time.sleep(pause * 1e-3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for i in range(40):
        gru._forward(x, y, False)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(40)]
    print("pause %5.1f ms: %s" % (pause, " ".join("%.0f" % (v * 1e3) for v in ms)))
