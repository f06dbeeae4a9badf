"""The multi-GPU entry points over the backend they run on in production: `nccl` (= RCCL on ROCm), one rank on this box's one GPU,
in a child process (the test process has its own HIP context; a process group wants a fresh one).  What N > 1 adds -- the partition
of the units and the order of the gathered results -- is covered on two gloo ranks in tests/test_shard_gloo.py; this checks that
the RCCL calls themselves (group creation bound to the device, gather of padded results, all-reduce and broadcast of a flat
gradient buffer, barrier) execute on the device and return what they should."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT
from tests.gpu_util import need_gpu

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import torch.distributed as dist
from sloika_amd import _lib, models, pipeline, shard, train
_lib.require_gpu()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=3))
bc = pipeline.Basecaller(net)
chunks = pipeline.synthetic_chunks(13, chunk_len=600, seed=9)
res = shard.basecall_sharded(lambda block: bc.call_chunks(torch.from_numpy(block).cuda()), chunks)
s, p, l = res
s0, p0, l0 = bc.call_chunks(torch.from_numpy(chunks).cuda())
assert torch.equal(s.cpu(), s0.cpu()) and torch.equal(p.cpu(), p0.cpu()) and torch.equal(l.cpu(), l0.cpu())
g = torch.arange(1000, dtype=torch.float32, device="cuda")
t = g.clone()
dist.all_reduce(t, op=dist.ReduceOp.SUM)                 # the collective of the training step, on the device
assert torch.equal(t, g)
assert train.allreduce_mean_(t) == 1.0 and torch.equal(train.broadcast_from_rank0_(t), g)
dist.barrier()
dist.destroy_process_group()
print("RCCL-OK")
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_sharded_basecall_and_gradient_collectives_on_rccl():
    need_gpu()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
