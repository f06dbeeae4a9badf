"""Time one training step (sloika_amd/train.py) on synthetic chunks.  Usage:
   python tools/bench_train.py [B] [T] [steps] [model]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sloika_amd import models, train  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
name = sys.argv[4] if len(sys.argv) > 4 else "raw_0.98_rgrgr"
net = models.randomise_zero_layers(models.build_model(name, klen=5, sd=0.5, seed=1))
fg = train.wrap_network(net, min_prob=1e-30, drop=20)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.normal(size=(T, B, net.insize)).astype(np.float32)).cuda()
To = net.layers[0].out_len(T) if hasattr(net.layers[0], "out_len") else T
labels = torch.from_numpy(rs.randint(0, net.size, size=(To, B)).astype(np.int32)).cuda()
weights = torch.ones((To, B), dtype=torch.float32, device="cuda")
for _ in range(2):
    loss, acc = fg(x, labels, weights, 1e-3)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    loss, acc = fg(x, labels, weights, 1e-3)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(name + " B=%d T=%d: %.2f ms per training step, %.1f M samples/s, loss %.4f acc %.4f, peak mem %.1f GB"
      % (B, T, dt * 1e3, B * T / dt / 1e6, loss, acc, torch.cuda.max_memory_allocated() / 2**30))
