import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
print({k: v for k, v in os.environ.items() if "HSA" in k or "HIP" in k or "ROC" in k or "GPU_" in k})
from sloika_amd import models
net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=11))
gru = net.layers[2]
x = torch.randn(8000, 1024, 96, device="cuda")   # a 5 ms kernel on every CU
y = torch.empty_like(x)
host = torch.empty(330 << 18, dtype=torch.float32).pin_memory()     # 346 MB
dev = torch.empty_like(host, device="cuda")
side = torch.cuda.Stream(priority=-1)
for busy in (False, True):
    for rep in range(3):
        torch.cuda.synchronize()
        if busy:
            for _ in range(6):
                gru._forward(x, y, False)
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            dev.copy_(host, non_blocking=True)
        side.synchronize()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("busy" if busy else "idle", "H2D 346 MB: %.1f ms (%.1f GB/s); device drained after %.1f ms" % ((t1 - t0) * 1e3, 0.346 / (t1 - t0), (t2 - t0) * 1e3))
