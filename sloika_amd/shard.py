"""Multi-GPU sharding of the path: reads / chunks are independent units (sloika's own parallelism is a process
pool over reads, sloika/iterators.py:343-351, bin/basecall_network.py:100-101), so rank r of N takes a contiguous
block of the chunks and NO collective sits on the data path.  The only communication is gathering the small
per-chunk results (paths, scores) to rank 0, over torch.distributed (RCCL on GPUs, gloo in the CPU tests).
"""
import numpy as np


def shard_bounds(n_units, rank, world_size):
    """Contiguous block [lo, hi) of `n_units` owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(n_units, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def dist_info():
    """(rank, world_size, local_rank) from the torchrun environment (defaults: single process)."""
    import os
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def gather_results(scores, paths, lens, n_units, group=None):
    """Gather per-chunk results of every rank to rank 0 in global chunk order.

    scores [n_local] float32, paths [n_local, T] int32, lens [n_local] int32 (torch tensors on the device the
    process group's backend expects).  Returns (scores, paths, lens) for all `n_units` on rank 0, None elsewhere.
    Ragged shards are padded to the largest shard for the collective and trimmed afterwards.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return scores, paths, lens
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(n_units, r, world)[1] - shard_bounds(n_units, r, world)[0] for r in range(world)]
    nmax, T = max(sizes), paths.shape[1]

    def pad(t, shape, fill):
        out = torch.full(shape, fill, dtype=t.dtype, device=t.device)
        out[: t.shape[0]] = t
        return out

    ps, pp, pl = pad(scores, (nmax,), 0), pad(paths, (nmax, T), -1), pad(lens, (nmax,), 0)
    if rank == 0:
        gs = [torch.empty_like(ps) for _ in range(world)]
        gp = [torch.empty_like(pp) for _ in range(world)]
        gl = [torch.empty_like(pl) for _ in range(world)]
    else:
        gs = gp = gl = None
    dist.gather(ps, gs, dst=0, group=group)
    dist.gather(pp, gp, dst=0, group=group)
    dist.gather(pl, gl, dst=0, group=group)
    if rank != 0:
        return None
    return (torch.cat([g[:n] for g, n in zip(gs, sizes)]), torch.cat([g[:n] for g, n in zip(gp, sizes)]),
            torch.cat([g[:n] for g, n in zip(gl, sizes)]))


def basecall_sharded(call_fn, chunks, group=None):
    """Run `call_fn(local_chunks) -> (scores, paths, lens)` on this rank's shard of `chunks` ([n, chunk_len],
    host array present on every rank or just indexable) and gather to rank 0."""
    import torch.distributed as dist
    n = len(chunks)
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(n, rank, world)
    scores, paths, lens = call_fn(chunks[lo:hi])
    return gather_results(scores, paths, lens, n, group)
