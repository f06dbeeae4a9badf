"""N > 1 path on CPU: two gloo ranks shard the chunks, each runs the decode stage on its block, results are
gathered to rank 0 in global order.  The compute function injected here is the CPU oracle (allowed in tests
only); on GPUs the same `shard.basecall_sharded` wraps `Basecaller.call_chunks`."""
import os
import socket
import sys

import numpy as np
import pytest

from tests.conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_units, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    from sloika_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, nst = 30, 65
    lp_all = np.log(np.random.RandomState(0).dirichlet(np.ones(nst) * 0.3, size=(n_units, T)).astype(np.float32) + 1e-6)

    def call(block):          # block: [n_local, T, nst]
        if len(block) == 0:
            return (torch.empty(0), torch.empty((0, T), dtype=torch.int32), torch.empty(0, dtype=torch.int32))
        s, p, l = orc.viterbi_batch(np.ascontiguousarray(np.transpose(block, (1, 0, 2))), 3, skip_pen=1.0)
        return torch.from_numpy(s), torch.from_numpy(p), torch.from_numpy(l)

    res = shard.basecall_sharded(call, lp_all)
    if rank == 0:
        s, p, l = res
        np.savez(os.path.join(out_dir, "gathered.npz"), s=s.numpy(), p=p.numpy(), l=l.numpy())
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [7, 8])
def test_two_rank_sharding_matches_single_process(tmp_path, n_units, oracle):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_units, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    T, nst = 30, 65
    lp_all = np.log(np.random.RandomState(0).dirichlet(np.ones(nst) * 0.3, size=(n_units, T)).astype(np.float32) + 1e-6)
    s, p, l = oracle.viterbi_batch(np.ascontiguousarray(np.transpose(lp_all, (1, 0, 2))), 3, skip_pen=1.0)
    assert np.array_equal(got["s"], s) and np.array_equal(got["p"], p) and np.array_equal(got["l"], l)
