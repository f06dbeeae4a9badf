"""Host logic of the training step (sloika_amd/train.py) that needs no GPU: which networks the step accepts, the ADAMski
step-size schedule (updates.py:54-76), and the data-parallel reduction on two gloo ranks -- each rank differentiates its
own half of the batch (here with the float64 oracle, allowed in tests only), one all-reduce + the returned scale must
reproduce the gradient of the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest

from tests.conftest import ROOT


def test_supported_architectures():
    from sloika_amd import layers, models, train
    body, sm = train._plan(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=1))
    assert [type(train._unwrap(l)[0]).__name__ for l in body] == ["Convolution"] + ["Gru"] * 5
    assert [train._unwrap(l)[1] for l in body] == [False, True, False, True, False, True] and isinstance(sm, layers.Softmax)
    train._plan(models.build_model("raw_1.00_rGr", klen=5, sd=0.5, seed=1))    # 110/142-wide layers: zero-padded
    train._plan(models.build_model("baseline_gru", klen=5, sd=0.5, seed=1))    # Window front end
    train._plan(models.build_model("tiny_gru", klen=5, sd=0.5, seed=1))
    train._plan(models.build_model("baseline_lstm", klen=5, sd=0.5, seed=1))   # peephole Lstm cells
    for name in ("baseline_raw_gru", "bigger_raw_gru"):                        # birnn (Parallel) + FeedForward stacks
        body, sm = train._plan(models.build_model(name, klen=5, sd=0.5, seed=1))
        kinds = set(type(l).__name__ for sub in body for l in train._leaves(sub))
        assert kinds == {"Convolution", "Gru", "FeedForward"}, kinds
    g = layers.Gru(4, 16)
    body, _ = train._plan(layers.Serial([layers.Reverse(layers.Reverse(g)), layers.Softmax(16, 5)]))
    assert train._unwrap(body[0]) == (g, False)
    for good in (layers.Serial([layers.Lstm(4, 8), layers.Softmax(8, 5)]),            # Lstm of 8: zero-padded to 16
                 layers.Serial([layers.Convolution(4, 8, 3), layers.Softmax(8, 5)]),   # multi-feature convolution
                 layers.Serial([layers.FeedForward(4, 1), layers.Convolution(1, 8, 3), layers.Softmax(8, 5)])):   # conv not first
        train._plan(good)
    for bad in (layers.Serial([g]),                                                      # no softmax
                layers.Serial([layers.Lstm(4, 130), layers.Softmax(130, 5)]),         # wider than the widest Lstm reverse scan
                layers.Serial([layers.Reverse(layers.Convolution(1, 8, 3)), layers.Softmax(8, 5)]),    # reversed convolution
                layers.Serial([layers.Gru(4, 150), layers.Softmax(150, 5)]),          # wider than the widest kernel
                layers.Serial([layers.Convolution(1, 8, 3), layers.Window(8, 3), layers.Softmax(24, 5)])):   # Window not first
        with pytest.raises(NotImplementedError):
            train._plan(bad)


@pytest.mark.parametrize("mrate", [0.0005, 0.01, None])
def test_adamski_schedule_matches_oracle(mrate):
    from oracle import oracle_train as ot
    from sloika_amd import train
    opt = ot.Adamski([np.zeros(1)], decay=(0.9, 0.999), mrate=mrate)
    t = 0.0
    for it in range(12):
        rate = 1e-3 / (1.0 + it / 5.0)
        lr, md, t = train.adamski_scalars(t, rate, (0.9, 0.999), mrate)
        want_lr, want_md = opt.scalars(rate)
        assert lr == float(want_lr) and md == float(want_md) and t == float(opt.t)


def test_helpers_mirror_reference():
    from sloika_amd import train
    labels = np.array([[3, 0, 0, 2, 0], [0, 0, 1, 0, 4]])
    assert train.remove_blanks(labels.copy()).tolist() == [[3, 3, 3, 2, 2], [0, 0, 1, 1, 4]]      # train_network.py:116-121
    sm = train.ExponentialSmoother(0.5)
    sm.update(2.0)
    sm.update(4.0)
    assert sm.value == pytest.approx((0.5 * 1.0 + 0.5 * 4.0) / (0.5 * 0.5 + 0.5), rel=1e-9)       # :100-113


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    rs = np.random.RandomState(4)
    n, nstate, T, B = 5, 7, 12, 6
    r = lambda *shape: rs.normal(size=shape) * 0.5
    spec = {"type": "serial", "sublayers": [
        {"type": "convolution", "W": r(n, 1, 3), "b": r(n), "stride": 1, "padding": (1, 1), "activation": "elu"},
        {"type": "reverse", "sublayer": {"type": "GRU", "iW": r(3 * n, n), "sW": r(2 * n, n), "sW2": r(n, n), "b": r(3 * n),
                                         "activation": "tanh", "gate": "sigmoid"}},
        {"type": "softmax", "W": r(nstate, n), "b": r(nstate)}]}
    x = rs.normal(size=(T, B, 1))
    labels = rs.randint(0, nstate, size=(T, B))
    weights = rs.uniform(0.5, 1.5, size=(T, B))
    return spec, x, labels, weights


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle_train as ot
    from sloika_amd import train
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec, x, labels, weights = _problem()
    mine = slice(rank, None, world)                                   # chunks rank::world, as the inference sharding
    loss, acc, grads = ot.loss_and_grads(spec, x[:, mine], labels[:, mine], weights[:, mine], 1e-3, 0.0, 1)
    flat = torch.from_numpy(np.concatenate([g.reshape(-1) for g in grads] + [[loss, acc]]))
    start = torch.full((4,), float(rank + 1))
    assert train.broadcast_from_rank0_(start).tolist() == [1.0] * 4      # replicas start from rank 0's parameters
    scale = train.allreduce_mean_(flat)
    if rank == 0:
        np.save(os.path.join(out_dir, "mean.npy"), flat.numpy() * scale)
    dist.barrier()
    dist.destroy_process_group()


def _flag_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from sloika_amd import train
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # [loss sum, accuracy sum, sum of squared parameters (the same on every rank), bad-label flag]: only rank 1 saw a bad label
    sc = torch.tensor([1.5 + rank, 10.0 * (rank + 1), 7.0, float(rank == 1)], dtype=torch.float64)
    train.allreduce_step_scalars_(sc)
    np.save(os.path.join(out_dir, "sc%d.npy" % rank), sc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_bad_label_flag_reaches_every_rank(tmp_path):
    """A label out of range on ONE rank must make EVERY rank raise before its update (the rank that saw it has already sent a garbage
    gradient into the all-reduce): the flag is summed over the ranks together with the loss and accuracy sums."""
    import torch
    import torch.multiprocessing as mp
    from sloika_amd import train
    sc = torch.tensor([1.0, 2.0, 3.0, 0.0], dtype=torch.float64)
    assert train.allreduce_step_scalars_(sc).tolist() == [1.0, 2.0, 3.0, 0.0]      # not initialised: a no-op
    port = _free_port()
    mp.spawn(_flag_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), "sc%d.npy" % rank))
        assert got.tolist() == [1.5 + 2.5, 10.0 + 20.0, 7.0, 1.0]                     # sums, the parameter term untouched, flag set


def test_two_rank_gradient_average_equals_whole_batch(tmp_path):
    import torch
    import torch.multiprocessing as mp
    from oracle import oracle_train as ot
    from sloika_amd import train
    assert train.allreduce_mean_(torch.ones(3)) == 1.0                 # not initialised: a no-op
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "mean.npy"))
    spec, x, labels, weights = _problem()
    loss, acc, grads = ot.loss_and_grads(spec, x, labels, weights, 1e-3, 0.0, 1)
    want = np.concatenate([g.reshape(-1) for g in grads] + [[loss, acc]])
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12)


def _toy_data(rs, n=40, chunk_len=120, stride=2, nstate=6):
    labels = rs.randint(0, nstate, size=(n, chunk_len // stride)).astype(np.int32)
    labels[rs.uniform(size=labels.shape) < 0.5] = 0
    bad = (rs.uniform(size=labels.shape) < 0.05).astype('i1')
    chunks = rs.normal(size=(n, chunk_len, 1)).astype(np.float32)
    weights = rs.uniform(size=n).astype(np.float32)
    weights[:3] = 0.0
    return chunks, labels, bad, weights


def test_chunk_file_preparation_and_sampler(tmp_path):
    """train_network.py:199-252 (data preparation) and :288-306 (the sampler), on a `.npz` chunk file."""
    from sloika_amd import train
    rs = np.random.RandomState(0)
    chunks, labels, bad, weights = _toy_data(rs)
    path = os.path.join(str(tmp_path), "chunks.npz")
    np.savez(path, chunks=chunks, labels=labels, bad=bad, weights=weights, kmer=np.int64(2), alphabet=np.bytes_(b"ACGT"))
    data = train.load_chunk_file(path)
    assert data["kmer"] == 2 and data["alphabet"] == b"ACGT" and data["weights"].dtype == np.float64
    all_labels, all_weights, label_weights = train.prepare_training_data(data, transducer=True, bad=True, ilf=False)
    assert all_weights.sum() == pytest.approx(1.0) and (all_labels[bad.astype(bool)] == 0).all()
    assert (all_labels[~bad.astype(bool)] == labels[~bad.astype(bool)]).all() and (label_weights == 1).all()
    lab2, _, lw = train.prepare_training_data(data, transducer=False, bad=False, ilf=True)
    assert (lab2[:, 1:] != 0).sum() >= (labels[:, 1:] != 0).sum() and lw.mean() == pytest.approx(1.0, rel=1e-6)
    for row, src in zip(lab2, labels):                       # remove_blanks: a blank repeats the label before it
        for j in range(1, len(row)):
            assert row[j] == (src[j] if src[j] != 0 else row[j - 1])
    # sampler: lengths multiple of the stride inside [min, max], label window aligned with the signal window, zero-weight
    # chunks never drawn, batch scaled inversely with the chunk length, learning rate decays, reproducible from the seed
    def draw(seed):
        np.random.seed(seed)
        return list(train.training_batches(data["chunks"], all_labels, all_weights, label_weights, niteration=25,
                                           batch_size=8, chunk_len_range=(0.5, 1.0), drop=5, rate=1e-3, lrdecay=10.0))
    a, b = draw(7), draw(7)
    assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, b))
    zero = set(np.flatnonzero(weights == 0))
    for i, (indata, lab, w, lr) in enumerate(a):
        clen, nb, _ = indata.shape
        assert 60 <= clen <= 120 and clen % 2 == 0 and lab.shape == (clen // 2, nb) and w.shape == lab.shape
        assert nb == min(int(8 * 120.0 / clen), 37)
        assert lr == pytest.approx(1e-3 / (1.0 + i / 10.0))
        # every drawn chunk is a window of one of the positive-weight chunks, with the matching label window
        for k in range(nb):
            hits = [(c, s) for c in range(40) if c not in zero for s in range(0, 120 - clen + 1, 2)
                    if np.array_equal(chunks[c, s:s + clen, 0], indata[:, k, 0])]
            assert len(hits) == 1
            c, s = hits[0]
            assert np.array_equal(all_labels[c, s // 2:(s + clen) // 2], lab[:, k])


def test_sampler_shares_a_batch_between_ranks():
    """Data-parallel sampling (train.training_batches with rank/world): identically seeded ranks draw the SAME global
    batch (window, length, chunk ids) and keep disjoint, equally sized shares of it."""
    from sloika_amd import train
    rs = np.random.RandomState(5)
    chunks, labels, bad, weights = _toy_data(rs)
    data = {"chunks": chunks, "labels": labels.copy(), "bad": bad, "weights": weights}
    all_labels, all_weights, label_weights = train.prepare_training_data(data)

    def batches(rank, world):
        np.random.seed(11)
        return list(train.training_batches(chunks, all_labels, all_weights, label_weights, 6, batch_size=9, drop=2,
                                           rank=rank, world=world))
    whole = batches(0, 1)
    for world in (2, 3):
        shares = [batches(r, world) for r in range(world)]
        for it in range(6):
            x_all, l_all, w_all, rate = whole[it]
            n = x_all.shape[1] - x_all.shape[1] % world
            parts = [shares[r][it] for r in range(world)]
            assert all(p[0].shape[0] == x_all.shape[0] and p[0].shape[1] == n // world for p in parts)   # same T, same B
            assert all(p[3] == rate for p in parts)
            for r, p in enumerate(parts):                   # rank r holds chunks r, r + world, ... of the global batch
                np.testing.assert_array_equal(p[0], x_all[:, :n][:, r::world])
                np.testing.assert_array_equal(p[1], l_all[:, :n][:, r::world])
    with pytest.raises(ValueError):
        np.random.seed(1)
        next(train.training_batches(chunks[:4], all_labels[:4], np.full(4, 0.25), label_weights, 1, batch_size=2, drop=2,
                                    rank=0, world=5))
