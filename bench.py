#!/usr/bin/env python3
"""Benchmark of the basecalling hot path on MI355X: raw-signal samples/s through
    normalise -> conv -> GRU stack -> softmax -> prepare_post+log -> k-mer Viterbi (+backtrace) -> paths on host.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model raw_0.98_rgrgr] [--batch 1024]

One "step" = one pass of the hot path over one batch of synthetic 4000-sample chunks already resident in HBM.
N > 1: launched by torch.distributed.run, one rank per GPU; every rank processes its own batch (reads/chunks are
independent units: no data-path collective, weak scaling); timing = max over ranks, value = whole-job samples/s.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for how `roofline` and `cpu_baseline` are defined).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# HIP maps streams onto 4 hardware queues by default; batches in flight on their own streams, each with side streams for the
# directions of a birnn and one for copies, serialise on them (baseline_raw_gru, B = 256, four in flight: 191 M samples/s
# with 4 queues, 345 M with 16; eight in flight: 322 M with 16, 417 M with 32).  Read by the HIP runtime when it starts, so set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_*_f32 = the fp32 vector rate
F16_MFMA_PEAK_TFLOPS = 2516.6     # dense fp16/bf16 MFMA = 16 x the fp32 rate (the ~2.5 PFLOP/s of the guide)
HBM_PEAK_GBS = 8000.0              # spec; ~6300 achievable

# GEMM-like flops per raw sample -- used for the end-to-end MFMA fraction
MFMA_STAGES = ("gru_fused", "gru_recurrent", "gru_input_gemm", "lstm_recurrent", "lstm_input_gemm", "softmax_gemm",
               "gemm_bias_act", "conv1d")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="raw_0.98_rgrgr")
    ap.add_argument("--batch", type=int, default=1024, help="chunks per GPU per step")
    ap.add_argument("--chunk-len", type=int, default=4000)
    ap.add_argument("--cpu-chunks", type=int, default=256,
                    help="chunks per slab of the CPU-baseline sample (slabs repeat until ~12 s; 0 = skip)")
    ap.add_argument("--no-stage-timing", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams per GPU; with 2, consecutive batches overlap (batch i decodes while batch i+1 "
                         "runs its recurrent layers)")
    ap.add_argument("--with-bases", action="store_true",
                    help="also turn the decoded paths into base sequences on the device (slk_paths_to_bases) and copy those "
                         "to the host, inside the timed step")
    ap.add_argument("--exact-steps", type=int, default=5,
                    help="steps of the all-fp32 arithmetic (SLOIKA_AMD_EXACT_F32=1) timed after the main region for the "
                         "`exact_f32` entry of the line (0 = skip)")
    ap.add_argument("--overlap-steps", type=int, default=10,
                    help="steps timed after the main region with TWO batches in flight on two streams, for the `two_in_flight` "
                         "entry of the line (0 = skip); the main region and `value` always use --streams (default 1)")
    ap.add_argument("--small-batch-steps", type=int, default=6,
                    help="steps of BASELINE.json configs[1] (baseline_raw_gru, batch 256) timed after the main region, one batch at "
                         "a time and eight in flight, for the `batch256` field (default workload on one GPU only; 0 = skip)")
    ap.add_argument("--train", action="store_true",
                    help="time the TRAINING step instead (BASELINE.json configs[4]: forward + backward + ADAMski, "
                         "gradient all-reduce over RCCL when --gpus > 1); prints the same kind of JSON line")
    return ap.parse_args()


def cpu_baseline(model_name, chunk_len, slab, budget_s=12.0, max_slabs=16):
    """The oracle (CPU port of the same pipeline: C + OpenMP over chunks) on the host cores of this box.
    Bounded sample: slabs of `slab` chunks of the same synthetic workload until ~budget_s seconds of CPU work."""
    from oracle import oracle as orc
    from sloika_amd import models, pipeline
    orc.build()
    net = models.randomise_zero_layers(models.build_model(model_name, klen=5, sd=0.5, seed=11))
    spec = net.spec()
    cores = orc.num_threads()
    done, spent = 0, 0.0
    for i in range(max_slabs):
        chunks = pipeline.synthetic_chunks(slab, chunk_len=chunk_len, seed=123, first_chunk=i * slab)
        t0 = time.perf_counter()
        x = orc.med_mad_normalise(chunks)
        post = orc.run_network(spec, np.ascontiguousarray(x.T)[:, :, None])
        lp = np.log(np.float32(1e-5) + np.float32(1.0 - 1e-5) * post + np.float32(1e-10))
        orc.viterbi_batch(lp, 5, skip_pen=0.0)
        spent += time.perf_counter() - t0
        done += slab
        del post, lp
        if spent >= budget_s:
            break
    out = {"value": done * chunk_len / spent, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": "%d chunks x %d samples of the same synthetic workload through the oracle C port "
                     "(OpenMP over chunks, %d threads), %.1f s" % (done, chunk_len, cores, spent)}
    # the same port on ONE core (SURVEY.md 8d asks for both): a dozen chunks, ~5 s
    try:
        orc.set_num_threads(1)
        n1 = 12
        chunks = pipeline.synthetic_chunks(n1, chunk_len=chunk_len, seed=123, first_chunk=0)
        t0 = time.perf_counter()
        x = orc.med_mad_normalise(chunks)
        post = orc.run_network(spec, np.ascontiguousarray(x.T)[:, :, None])
        lp = np.log(np.float32(1e-5) + np.float32(1.0 - 1e-5) * post + np.float32(1e-10))
        orc.viterbi_batch(lp, 5, skip_pen=0.0)
        t1 = time.perf_counter() - t0
        out["single_core"] = {"value": n1 * chunk_len / t1, "unit": "samples/s",
                              "sample": "%d chunks, 1 thread, %.1f s" % (n1, t1)}
    finally:
        orc.set_num_threads(cores)
    return out


def cpu_baseline_train(model_name, chunk_len, nchunk=48):
    """The training oracle (oracle/oracle_train.py: numpy float64 forward + hand-derived reverse pass + ADAMski) on a bounded
    sample of the same workload, on the host: the CPU restatement of one fg(x, labels, weights, rate) call."""
    from oracle import oracle_train as ot
    from sloika_amd import models
    net = models.randomise_zero_layers(models.build_model(model_name, klen=5, sd=0.5, seed=11))
    spec = net.spec()
    rs = np.random.RandomState(99)
    To = net.layers[0].out_len(chunk_len) if hasattr(net.layers[0], "out_len") else chunk_len
    x = rs.normal(size=(chunk_len, nchunk, net.insize)).astype(np.float32)
    labels = rs.randint(0, net.size, size=(To, nchunk)).astype(np.int32)
    weights = np.ones((To, nchunk), dtype=np.float32)
    t0 = time.perf_counter()
    loss, acc, grads = ot.loss_and_grads(spec, x, labels, weights, 1e-30, 0.0, 20)
    params = ot.params_of(spec)
    ot.Adamski(params).step(params, grads, 1e-3)
    dt = time.perf_counter() - t0
    try:
        import threadpoolctl
        cores = max(int(i.get("num_threads", 1)) for i in threadpoolctl.threadpool_info()) if threadpoolctl.threadpool_info() else 1
    except Exception:
        cores = 1
    return {"value": nchunk * chunk_len / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%d chunks x %d samples, one forward + backward + ADAMski step of the numpy float64 training oracle "
                      "(BLAS threads: %d), %.1f s" % (nchunk, chunk_len, cores, dt)}


def main_train(args):
    """One step = wrap_network's fg(x, labels, weights, rate) on one batch per GPU (bin/train_network.py:308)."""
    import torch
    from sloika_amd import _lib, models, profiler, shard, train
    rank, world, local_rank = shard.dist_info()
    _lib.require_gpu()
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    dist = None
    if "WORLD_SIZE" in os.environ:          # under a launcher (also with one rank: the RCCL path is the path that runs)
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    net = models.randomise_zero_layers(models.build_model(args.model, klen=5, sd=0.5, seed=11))     # same weights on every rank
    fg = train.wrap_network(net, min_prob=1e-30, l2=0.0, drop=20)                                  # train_network.py defaults
    B, L = args.batch, args.chunk_len
    rs = np.random.RandomState(1234 + rank)
    To = net.layers[0].out_len(L) if hasattr(net.layers[0], "out_len") else L        # event-feature models keep the length
    x = torch.from_numpy(rs.normal(size=(L, B, net.insize)).astype(np.float32)).cuda()
    labels = torch.from_numpy(rs.randint(0, net.size, size=(To, B)).astype(np.int32)).cuda()
    weights = torch.ones((To, B), dtype=torch.float32, device="cuda")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        fg(x, labels, weights, 1e-3)
    barrier()
    rec = None if args.no_stage_timing else profiler.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, acc = fg(x, labels, weights, 1e-3 / (1.0 + i / 5000.0))
    barrier()
    dt = time.perf_counter() - t0
    if rec is not None:
        profiler.stop()
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stages = rec.summary() if rec is not None else {}
    roofline = None
    # HBM bytes per step of every stage from the committed PMC passes of this same workload (tools/collect_pmc.sh --train)
    stage_traffic = {}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_train.json")) as fh:
            pmc = json.load(fh)
        if pmc.get("workload") == [args.model, args.batch, args.chunk_len]:
            stage_traffic = pmc.get("stage_bytes_per_step", {})
    except (OSError, ValueError):
        pass
    if stages:
        dom = max(stages, key=lambda k: stages[k]["ms_total"])
        d = stages[dom]
        flops = d["flops"] / d["calls"]
        f16 = d.get("f16x3_flops", 0.0) / d["calls"]
        t_min = (flops - f16) / (FP32_MFMA_PEAK_TFLOPS * 1e12) + 3.0 * f16 / (F16_MFMA_PEAK_TFLOPS * 1e12)
        ach = flops / (d["ms_avg"] * 1e-3) / 1e12
        per_launch = stage_traffic[dom] * args.steps / d["calls"] if dom in stage_traffic else None
        roofline = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": flops / t_min / 1e12, "unit": "TFLOP/s",
                    "frac": ach / (flops / t_min / 1e12), "traffic": per_launch, "ms_per_launch": d["ms_avg"],
                    "launches": d["calls"]}
    if rank == 0:
        print(json.dumps({
            "metric": "raw-signal samples/sec trained", "value": world * B * L * args.steps / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (forward products and the weight-gradient-free GEMMs as 3-term fp16 splits, f32 accumulation)",
            "data": "synthetic",
            "config": {"workload": "%s training step (forward, backward, ADAMski), %d-sample chunks, batch %d per GPU, "
                                   "klen 5 (1025 states), drop 20" % (args.model, L, B),
                       "model": args.model, "chunk_len": L, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": "data parallel over %d GPU(s), one all-reduce of the flat gradient per step" % world},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline_train(args.model, L) if (world == 1 and args.cpu_chunks > 0) else None,
            "final_loss": loss,
            "stages_ms_per_step": {k: v["ms_total"] / args.steps for k, v in sorted(stages.items())},
            "stages_hbm_bytes_per_step": {k: stage_traffic[k] for k in sorted(stages) if k in stage_traffic} or None}))
    if dist is not None:
        dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script under torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1) as a CHILD process and exit with its return code.  Nothing here has touched the GPU
    (no torch.cuda call, no library load), and the parent never replaces itself with another program."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if world_env is not None and int(world_env) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks\n" % (args.gpus, world_env))
        sys.exit(2)
    sys.stderr.write("bench.py: rank %s of %s starting\n" % (os.environ.get("RANK", "0"), world_env or "1"))
    sys.stderr.flush()
    if args.train:
        return main_train(args)
    import torch
    from sloika_amd import _lib, models, pipeline, profiler, shard
    rank, world, local_rank = shard.dist_info()
    _lib.require_gpu()
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    dist = None
    if "WORLD_SIZE" in os.environ:          # under a launcher (also with one rank: the RCCL path is the path that runs)
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    net = models.randomise_zero_layers(models.build_model(args.model, klen=5, sd=0.5, seed=11))
    nstream = max(1, args.streams)
    nslot = max(nstream, 4 if (args.overlap_steps > 0 and nstream == 1) else 1)     # the in-flight legs need up to four slots
    bcs = [pipeline.Basecaller(net, kmer_len=5, nbase=4, min_prob=1e-5, skip=0.0, in_flight=nstream) for _ in range(nslot)]
    bc = bcs[0]
    streams = ([torch.cuda.Stream() for _ in range(nslot)] if nstream > 1
               else [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nslot - 1)])
    B, L = args.batch, args.chunk_len
    # a few distinct batches so that steps do not all hit the same cache lines
    nbuf = 2
    host = [pipeline.synthetic_chunks(B, chunk_len=L, seed=0xdeadbeef, first_chunk=(rank * nbuf + i) * B)
            for i in range(nbuf)]
    dev = [torch.from_numpy(h).cuda() for h in host]
    tout = bc.network.layers[0].out_len(L) if hasattr(bc.network.layers[0], "out_len") else L
    out_host = [torch.empty((B, tout), dtype=torch.int32).pin_memory() for _ in range(nslot)]
    klen = 5
    if args.with_bases:
        bases_dev = [torch.empty((B, klen * tout), dtype=torch.uint8, device="cuda") for _ in range(nslot)]
        nbases_dev = [torch.empty((B,), dtype=torch.int32, device="cuda") for _ in range(nslot)]
        bases_host = [torch.empty((B, klen * tout), dtype=torch.uint8).pin_memory() for _ in range(nslot)]
        nbases_host = [torch.empty((B,), dtype=torch.int32).pin_memory() for _ in range(nslot)]
        acgt = int.from_bytes(b"ACGT".ljust(8, b"\0"), "little")

    # The results leave for the host on a copy stream of their own: the next step's kernels do not queue behind a PCIe
    # transfer.  `paths` is a fresh tensor every step (record_stream keeps the allocator from reusing it before the copy
    # has run); the persistent base buffers are protected by the copy's event.
    copy_stream = torch.cuda.Stream()
    copied = [None] * nslot

    def step(i, nact=nstream):
        k = i % nact
        with torch.cuda.stream(streams[k]):
            scores, paths, lens = bcs[k].call_chunks(dev[i % nbuf])
            paths.record_stream(copy_stream)
            if args.with_bases:                                                 # base sequences, still on the device
                if copied[k] is not None:
                    streams[k].wait_event(copied[k])
                with profiler.region("bases", 0.0, 5.0 * paths.numel()):
                    _lib.check(_lib.lib().slk_paths_to_bases(paths.data_ptr(), paths.stride(0), lens.data_ptr(), B, klen, 4, 1,
                                                             acgt, bases_dev[k].data_ptr(), klen * tout, nbases_dev[k].data_ptr(),
                                                             streams[k].cuda_stream), "paths_to_bases")
            done = torch.cuda.Event()
            done.record(streams[k])
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(done)
            out_host[k][:, : paths.shape[1]].copy_(paths, non_blocking=True)     # paths end up on the host
            if args.with_bases:                                                 # ... and so do the base sequences
                bases_host[k].copy_(bases_dev[k], non_blocking=True)
                nbases_host[k].copy_(nbases_dev[k], non_blocking=True)
            copied[k] = torch.cuda.Event()
            copied[k].record(copy_stream)
        return scores, lens

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    rec = None if args.no_stage_timing else profiler.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if rec is not None:
        profiler.stop()
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    samples = world * B * L * args.steps
    value = samples / dt

    # the same workload with every product in plain float32 MFMA (no fp16 splits anywhere): a few steps, same run
    exact = None
    if args.exact_steps > 0 and nstream == 1:
        from sloika_amd import layers as _layers
        keep = (_layers.SPLIT_F16, _layers.Softmax.split_f16)
        _layers.SPLIT_F16, _layers.Softmax.split_f16 = False, False
        try:
            step(0)
            barrier()
            t1 = time.perf_counter()
            for i in range(args.exact_steps):
                step(i)
            barrier()
            dte = time.perf_counter() - t1
            if dist is not None:
                tm = torch.tensor([dte], dtype=torch.float64, device="cuda")
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dte = float(tm.item())
            exact = {"ms_per_step": dte / args.exact_steps * 1e3, "value": world * B * L * args.exact_steps / dte,
                     "unit": "samples/s", "steps": args.exact_steps,
                     "arithmetic": "float32 MFMA for every product (SLOIKA_AMD_EXACT_F32=1): two-kernel Gru, fp32 softmax projection"}
        finally:
            _layers.SPLIT_F16, _layers.Softmax.split_f16 = keep

    # the same workload with two batches in flight (two streams, Basecaller(in_flight=2)): batch i decodes while batch i+1 runs
    # its recurrent layers, and two batches' recurrent layers share the chip on the eight-chunk plan
    overlap = None
    if args.overlap_steps > 0 and nstream == 1:
        for bco in bcs:                       # every Basecaller is told that two batches are in flight (eight-chunk Gru plan)
            bco.in_flight = 2
        for i in range(4):                    # the second slot allocates its buffers on first use
            step(i, 2)
        barrier()
        t2 = time.perf_counter()
        for i in range(args.overlap_steps):
            step(i, 2)
        barrier()
        dto = time.perf_counter() - t2
        if dist is not None:
            tm = torch.tensor([dto], dtype=torch.float64, device="cuda")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dto = float(tm.item())
        overlap = {"ms_per_step": dto / args.overlap_steps * 1e3, "value": world * B * L * args.overlap_steps / dto,
                   "unit": "samples/s", "steps": args.overlap_steps, "streams_per_gpu": 2}
        # ... four in flight on four streams: each batch's recurrent layers take a quarter of the chip on the sixteen-chunk plan
        for bco in bcs:
            bco.in_flight = 4
        for i in range(8):
            step(i, 4)
        barrier()
        t6 = time.perf_counter()
        n4 = 2 * args.overlap_steps
        for i in range(n4):
            step(i, 4)
        barrier()
        dt4 = time.perf_counter() - t6
        if dist is not None:
            tm = torch.tensor([dt4], dtype=torch.float64, device="cuda")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt4 = float(tm.item())
        overlap["four_in_flight"] = {"ms_per_step": dt4 / n4 * 1e3, "value": world * B * L * n4 / dt4, "unit": "samples/s",
                                     "steps": n4, "streams_per_gpu": 4}
        for bco in bcs:
            bco.in_flight = nstream
        # ... and with the two batches handed over as ONE call of 2B chunks: a recurrent layer then runs the eight-chunk plan
        # (csrc/gru_bar16d.hip: one workgroup per CU takes a 4-chunk tile of EACH batch through the same MFMAs) instead of two
        # rounds of four-chunk workgroups.  Only the paths of the first B chunks are copied out per B chunks of work, as above.
        if not args.with_bases:
            pair = torch.cat([dev[0], dev[1 % nbuf]], dim=0)
            out2 = torch.empty((2 * B, tout), dtype=torch.int32).pin_memory()
            def step_pair():
                scores, paths, lens = bc.call_chunks(pair)
                paths.record_stream(copy_stream)
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream())
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(done)
                    out2[:, : paths.shape[1]].copy_(paths, non_blocking=True)
            npair = max(1, args.overlap_steps // 2)
            for _ in range(2):
                step_pair()
            barrier()
            t3 = time.perf_counter()
            for _ in range(npair):
                step_pair()
            barrier()
            dtp = time.perf_counter() - t3
            if dist is not None:
                tm = torch.tensor([dtp], dtype=torch.float64, device="cuda")
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dtp = float(tm.item())
            overlap["as_one_batch"] = {"ms_per_step": dtp / (2 * npair) * 1e3, "value": world * 2 * B * L * npair / dtp,
                                       "unit": "samples/s", "steps": 2 * npair, "chunks_per_call": 2 * B,
                                       "note": "ms_per_step is per %d chunks; one call carries two batches" % B}
            del pair
            # ... and four batches as one call of 4B chunks: the recurrent layers run sixteen chunks per workgroup
            # (csrc/gru_bar16q.hip: every column of the recurrent MFMAs a different chunk)
            quad = torch.cat([dev[i % nbuf] for i in range(4)], dim=0)
            out4 = torch.empty((4 * B, tout), dtype=torch.int32).pin_memory()
            def step_quad():
                scores, paths, lens = bc.call_chunks(quad)
                paths.record_stream(copy_stream)
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream())
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(done)
                    out4[:, : paths.shape[1]].copy_(paths, non_blocking=True)
            nquad = max(1, args.overlap_steps // 4)
            step_quad()
            barrier()
            t5 = time.perf_counter()
            for _ in range(nquad):
                step_quad()
            barrier()
            dtq = time.perf_counter() - t5
            if dist is not None:
                tm = torch.tensor([dtq], dtype=torch.float64, device="cuda")
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dtq = float(tm.item())
            overlap["four_as_one_batch"] = {"ms_per_step": dtq / (4 * nquad) * 1e3, "value": world * 4 * B * L * nquad / dtq,
                                            "unit": "samples/s", "steps": 4 * nquad, "chunks_per_call": 4 * B,
                                            "note": "ms_per_step is per %d chunks; one call carries four batches" % B}
            del quad

    # BASELINE.json configs[1] -- the batch north_star quotes (baseline_raw_gru, 256 chunks of 4000 samples): every stage is
    # latency bound at that size (64 workgroups per recurrent launch, 256 decoder workgroups), so the device only fills up with
    # several batches in flight
    small = None
    if (args.small_batch_steps > 0 and world == 1 and nstream == 1 and args.model == "raw_0.98_rgrgr" and args.batch == 1024
            and not args.with_bases):
        net1 = models.randomise_zero_layers(models.build_model("baseline_raw_gru", klen=5, sd=0.5, seed=11))
        B1, nfl = 256, 8
        bcs1 = [pipeline.Basecaller(net1, kmer_len=5, nbase=4, min_prob=1e-5, skip=0.0) for _ in range(nfl)]
        st1 = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nfl - 1)]
        dev1 = [torch.from_numpy(pipeline.synthetic_chunks(B1, chunk_len=L, seed=0xfeed, first_chunk=i * B1)).cuda() for i in range(nfl)]
        tout1 = bcs1[0].network.layers[0].out_len(L)
        host1 = [torch.empty((B1, tout1), dtype=torch.int32).pin_memory() for _ in range(nfl)]

        def step1(i, nact):
            k = i % nact
            with torch.cuda.stream(st1[k]):
                scores, paths, lens = bcs1[k].call_chunks(dev1[k])
                paths.record_stream(copy_stream)
                done = torch.cuda.Event()
                done.record(st1[k])
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done)
                host1[k][:, : paths.shape[1]].copy_(paths, non_blocking=True)

        small = {"workload": "baseline_raw_gru inference, %d-sample chunks, batch %d (BASELINE.json configs[1])" % (L, B1)}
        for nact, key in ((1, "one_at_a_time"), (nfl, "eight_in_flight")):
            for bco in bcs1:
                bco.in_flight = nact
            for i in range(2 * nact):
                step1(i, nact)
            barrier()
            t4 = time.perf_counter()
            nst = args.small_batch_steps * nact
            for i in range(nst):
                step1(i, nact)
            barrier()
            d4 = time.perf_counter() - t4
            small[key] = {"ms_per_step": d4 / nst * 1e3, "value": B1 * L * nst / d4, "unit": "samples/s", "steps": nst,
                          "streams_per_gpu": nact}
        del bcs1, dev1

    stages = rec.summary() if rec is not None else {}
    roofline = None
    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload (tools/collect_pmc.sh);
    # only quoted when the file describes the configuration being run.
    traffic_by_stage = {}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        if pmc.get("workload") == [args.model, args.batch, args.chunk_len]:
            traffic_by_stage = pmc.get("stage_bytes_per_launch", {})
    except (OSError, ValueError):
        pass
    if stages:
        dom = max(stages, key=lambda k: stages[k]["ms_total"])
        d = stages[dom]
        if dom in MFMA_STAGES:
            # The matrix roofline of this launch for ITS instruction mix: the fp32 part at the fp32 MFMA peak, the part
            # evaluated as a 3-term fp16 split (three fp16 MFMAs per product) at the fp16 peak.  `peak` is the
            # algorithmic FLOP/s the kernel would reach with both pipes saturated, so frac = t_roofline / t_measured.
            flops = d["flops"] / d["calls"]
            f16 = d.get("f16x3_flops", 0.0) / d["calls"]
            t_min = (flops - f16) / (FP32_MFMA_PEAK_TFLOPS * 1e12) + 3.0 * f16 / (F16_MFMA_PEAK_TFLOPS * 1e12)
            ach = flops / (d["ms_avg"] * 1e-3) / 1e12
            peak = flops / t_min / 1e12
            roofline = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak,
                        "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic_by_stage.get(dom),
                        "ms_per_launch": d["ms_avg"], "launches": d["calls"],
                        "mix": {"fp32_mfma_flops": flops - f16, "f16x3_flops": f16,
                                "fp32_peak": FP32_MFMA_PEAK_TFLOPS, "f16_peak": F16_MFMA_PEAK_TFLOPS},
                        # SURVEY 8(d)'s yardstick for the NN stage (all flops at the fp32 MFMA peak), for comparison only
                        "frac_vs_fp32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS,
                        # B = 1024 over 256 CUs leaves 4 chunks per workgroup: 4 of the 16 columns of a 16x16x32 MFMA tile
                        "note": "latency-bound serial scan; 4 chunks per CU fill 4 of 16 MFMA columns"}
        else:
            ach = d["bytes"] / d["calls"] / (d["ms_avg"] * 1e-3) / 1e9
            roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": traffic_by_stage.get(dom), "ms_per_launch": d["ms_avg"],
                        "launches": d["calls"]}
    if rank == 0:
        cpu = None
        if world == 1 and args.cpu_chunks > 0:
            cpu = cpu_baseline(args.model, L, args.cpu_chunks)
        gemm_flops = sum(stages[k]["flops"] for k in stages if k in MFMA_STAGES) / max(1, args.steps)
        line = {
            "metric": "raw-signal samples/sec basecalled", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (products as 3-term fp16 splits with f32 accumulation, 22-bit operands; elementwise and DP in f32)",
            "data": "synthetic",
            "config": {"workload": "%s inference, %d-sample chunks, batch %d per GPU, klen 5 (1025 states), "
                                   "normalise->conv->GRU->softmax->Viterbi->paths%s on host; matrix products as fp16x3 splits "
                                   "(f32-grade, see exact_f32 for plain fp32 MFMA)"
                                   % (args.model, L, B, " + base sequences" if args.with_bases else ""),
                       "model": args.model, "chunk_len": L, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": "chunks sharded over %d GPU(s), no collective" % world, "streams_per_gpu": nstream},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "exact_f32": exact,
            "two_in_flight": overlap,
            "batch256": small,
            "stages_ms_per_step": {k: v["ms_total"] / args.steps for k, v in sorted(stages.items())},
            "e2e_algorithmic_tflops": (gemm_flops / (dt / args.steps) / 1e12) if stages else None,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
