#!/usr/bin/env python3
"""Benchmark of the basecalling hot path on MI355X: raw-signal samples/s through
    normalise -> conv -> GRU stack -> softmax -> prepare_post+log -> k-mer Viterbi (+backtrace) -> paths on host.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model raw_0.98_rgrgr] [--batch 1024]

One "step" = one pass of the hot path over one batch of synthetic 4000-sample chunks already resident in HBM.
N > 1: launched by torch.distributed.run, one rank per GPU; every rank processes its own batch (reads/chunks are
independent units: no data-path collective, weak scaling); timing = max over ranks, value = whole-job samples/s.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for how `roofline` and `cpu_baseline` are defined).  Besides the
contract's fields the line carries, all measured in the same process on the same device:
    roofline            dominant kernel of the workload: algorithmic flops / HIP-event duration against ITS pipe's peak
    exact_f32           the same workload with every product in plain fp32 MFMA
    in_flight           two / four batches in flight on streams of their own, two / four batches as one call
    batch256            BASELINE.json configs[1] (baseline_raw_gru) AND the metric's model (raw_0.98_rgrgr) at the batch the
                        north star quotes, one batch at a time and eight in flight, each with its own roofline
    with_upload         the step including the PCIe upload of the next batch's raw signal (copy stream)
    sustained           >= 10 s of back-to-back steps, one batch at a time and four in flight, with the shader clock sampled
    whole_reads         the reference's own inference mode (whole reads, basecall.py:88-121): reads of 50k-115k samples,
                        bucketed by length, ragged batches in flight
    train               a few steps of the training step (BASELINE.json configs[4] on this GPU)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# HIP maps streams onto 4 hardware queues by default; batches in flight on their own streams serialise on them (baseline_raw_gru,
# B = 256, eight in flight: 191 M samples/s with 4 queues, 417 M with 32).  Read by the HIP runtime when it starts; importing
# sloika_amd sets the same default.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_*_f32 = the fp32 vector rate
F16_MFMA_PEAK_TFLOPS = 2516.6     # dense fp16/bf16 MFMA = 16 x the fp32 rate (the ~2.5 PFLOP/s of the guide)
HBM_PEAK_GBS = 8000.0              # spec; ~6300 achievable

# stages whose work is matrix products (priced against the MFMA peaks); the others against HBM
MFMA_STAGES = ("gru_fused", "gru_recurrent", "gru_input_gemm", "lstm_fused", "lstm_recurrent", "lstm_input_gemm", "softmax_gemm",
               "gemm_bias_act", "conv1d", "softmax_viterbi",
               # the training step's scans and softmax layer (its weight-gradient and dL/dx contractions are priced against HBM:
               # TRAIN_HBM_STAGES)
               "train_gru_scan", "train_lstm_scan", "train_softmax_xent", "train_gates", "train_wgrad", "train_dx")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step allocates nothing (Basecaller(borrow=True): every buffer out of the Basecaller's own arena), so a handful of warm-up steps
    # is enough: the first call sizes the arena and builds the weight packs, the rest run like the thousandth (up to round 5 every layer
    # asked torch's caching allocator for its output and a fresh process needed ~100 steps to settle: 4.14-4.21 ms behind 10 warm-up
    # steps against 3.91 behind 200)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="raw_0.98_rgrgr")
    ap.add_argument("--batch", type=int, default=1024, help="chunks per GPU per step")
    ap.add_argument("--chunk-len", type=int, default=4000)
    ap.add_argument("--cpu-chunks", type=int, default=256,
                    help="chunks per slab of the CPU-baseline sample (slabs repeat until ~12 s; 0 = skip)")
    ap.add_argument("--no-stage-timing", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams per GPU for the MAIN region; with 2, consecutive batches overlap")
    ap.add_argument("--with-bases", action="store_true",
                    help="also turn the decoded paths into base sequences on the device (slk_paths_to_bases) and copy those "
                         "to the host, inside the timed step")
    ap.add_argument("--stage-steps", type=int, default=10,
                    help="steps of the separate pass with per-stage HIP events (roofline, stages_ms_per_step); the timed region "
                         "itself carries no events")
    ap.add_argument("--exact-steps", type=int, default=5, help="steps of the all-fp32 arithmetic for `exact_f32` (0 = skip)")
    ap.add_argument("--overlap-steps", type=int, default=10, help="steps of the in-flight legs (0 = skip)")
    ap.add_argument("--small-batch-steps", type=int, default=6, help="steps of the batch-256 legs (0 = skip)")
    ap.add_argument("--sustained-seconds", type=float, default=10.0, help="length of each sustained leg (0 = skip)")
    ap.add_argument("--upload-steps", type=int, default=10, help="steps of the leg that uploads the next batch (0 = skip)")
    # 4096: with 1024 reads the longest read's own chain (115 000 samples through five sequential layers) takes longer than the whole
    # set would at the chunk-mode rate -- no schedule can do better than 0.68 x -- so a set that small measures the read, not the path
    ap.add_argument("--whole-reads", type=int, default=4096, help="synthetic whole reads of 50k-115k samples (0 = skip)")
    ap.add_argument("--train-steps", type=int, default=10, help="training steps for the `train` field (0 = skip)")
    ap.add_argument("--host-feed-steps", type=int, default=10,
                    help="steps of the leg that uploads EIGHT ranks' raw signal per step from pinned host memory (0 = skip)")
    ap.add_argument("--stub-device", action="store_true",
                    help="tests only (tests/test_bench_launcher.py): the rank plumbing of this script -- device binding by LOCAL_RANK, "
                         "barriers, max over ranks, one line from rank 0 -- on the gloo backend with a Runner that sleeps; no GPU, "
                         "no kernel, and the line says so")
    ap.add_argument("--quick", action="store_true", help="only the main region, the stage pass and the CPU baseline")
    ap.add_argument("--only", default=None, choices=["batch256", "model1024"],
                    help="run ONE leg and print its JSON (bench.py starts itself with this as a child process)")
    ap.add_argument("--train", action="store_true",
                    help="time the TRAINING step instead (BASELINE.json configs[4]: forward + backward + ADAMski, "
                         "gradient all-reduce over RCCL when --gpus > 1); prints the same kind of JSON line")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
# CPU baselines (the oracle: test infrastructure, timed here as the reported CPU column)
# ----------------------------------------------------------------------------------------------------------------------
def feature_input(L, B, nfeature, seed):
    return np.random.RandomState(1000 + seed).normal(size=(L, B, nfeature)).astype(np.float32)


def cpu_baseline(model_name, chunk_len, slab, budget_s=12.0, max_slabs=16):
    """The oracle (CPU port of the same pipeline: C + OpenMP over chunks) on the host cores of this box.
    Bounded sample: slabs of `slab` chunks of the same synthetic workload until ~budget_s seconds of CPU work."""
    from oracle import oracle as orc
    from sloika_amd import models, pipeline
    orc.build()
    net = models.randomise_zero_layers(models.build_model(model_name, klen=5, sd=0.5, seed=11))
    spec = net.spec()
    # the baseline runs the build made for THIS box's cores (-O3 -march=native, oracle/Makefile `native`; same arithmetic flags), once it
    # has reproduced the portable build's bits on a few chunks; the portable build (-O2 -march=x86-64-v2) otherwise
    build_kind = "portable (-O2 -march=x86-64-v2)"
    if net.insize == 1:
        probe = pipeline.synthetic_chunks(4, chunk_len=min(chunk_len, 500), seed=77)
        xin = np.ascontiguousarray(orc.med_mad_normalise(probe).T)[:, :, None]
        want = orc.run_network(spec, xin)
        native = orc.build_native()
        if native is not None:
            orc.use_library(native)
            try:
                same = np.array_equal(orc.run_network(spec, xin), want)
            except OSError:
                same = False
            if same:
                build_kind = "native (-O3 -march=native, bits equal to the portable build's)"
            else:
                orc.use_library(None)
    cores = orc.num_threads()
    # every thread gets the same number of chunks (static OpenMP schedule over chunks): a slab is a multiple of the thread count, at
    # least four chunks per thread -- round 4's 256-chunk slabs gave 128 threads two chunks each and left the logarithm of the posteriors
    # (840 MB per slab) to ONE numpy thread: 9.3 x one core was mostly that
    slab = max(slab, 4 * cores) // cores * cores if cores > 1 else slab

    def log_post(post):
        """prepare_post + log (decode.py:36, :56) over the chunks on all cores (numpy releases the GIL inside the ufuncs)."""
        import concurrent.futures
        out = np.empty_like(post)
        nb = post.shape[1]
        cuts = [(k * nb // cores, (k + 1) * nb // cores) for k in range(cores)]

        def one(c):
            lo, hi = c
            if hi > lo:
                np.log(np.float32(1e-5) + np.float32(1.0 - 1e-5) * post[:, lo:hi] + np.float32(1e-10), out=out[:, lo:hi])
        with concurrent.futures.ThreadPoolExecutor(max_workers=cores) as ex:
            list(ex.map(one, cuts))
        return out

    done, spent = 0, 0.0
    for i in range(max_slabs):
        feats = net.insize != 1
        chunks = feature_input(chunk_len, slab, net.insize, 7000 + i) if feats else \
            pipeline.synthetic_chunks(slab, chunk_len=chunk_len, seed=123, first_chunk=i * slab)
        t0 = time.perf_counter()
        post = orc.run_network(spec, chunks if feats else np.ascontiguousarray(orc.med_mad_normalise(chunks).T)[:, :, None])
        lp = log_post(post)
        orc.viterbi_batch(lp, 5, skip_pen=0.0)
        spent += time.perf_counter() - t0
        done += slab
        del post, lp
        if spent >= budget_s:
            break
    out = {"value": done * chunk_len / spent, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": "%d chunks x %d samples of the same synthetic workload through the oracle C port "
                     "(OpenMP over chunks, %d threads), %.1f s" % (done, chunk_len, cores, spent), "build": build_kind}
    # the same port on ONE core (SURVEY.md 8d asks for both): a dozen chunks, ~5 s
    try:
        orc.set_num_threads(1)
        n1 = 12
        chunks = feature_input(chunk_len, n1, net.insize, 7000) if feats else \
            pipeline.synthetic_chunks(n1, chunk_len=chunk_len, seed=123, first_chunk=0)
        t0 = time.perf_counter()
        post = orc.run_network(spec, chunks if feats else np.ascontiguousarray(orc.med_mad_normalise(chunks).T)[:, :, None])
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


# ----------------------------------------------------------------------------------------------------------------------
# helpers shared by the legs
# ----------------------------------------------------------------------------------------------------------------------
def price_stage(name, d, traffic=None, hbm_stage=False):
    """One stage of a pass timed with HIP events against the peak of the pipe it runs on: algorithmic work per launch / average
    launch duration.  Matrix stages: the fp32 part at the fp32 MFMA peak, the part evaluated as a 3-term fp16 split at three fp16
    MFMAs per product, the part whose lo half rides in spare MFMA columns (or that is split in two terms) at two -- i.e. what the
    kernel really issues (`mix`); the others, and the stages named in TRAIN_HBM_STAGES, against HBM on their ALGORITHMIC bytes.
    `traffic` = HBM bytes per launch by the PMC counters (or None)."""
    calls = d["calls"]
    flops = d["flops"] / calls
    if name in MFMA_STAGES and flops > 0 and not hbm_stage:
        f16 = d.get("f16x3_flops", 0.0) / calls
        f16x2 = d.get("f16x2_flops", 0.0) / calls
        t_min = (flops - f16 - f16x2) / (FP32_MFMA_PEAK_TFLOPS * 1e12) + (3.0 * f16 + 2.0 * f16x2) / (F16_MFMA_PEAK_TFLOPS * 1e12)
        ach = flops / (d["ms_avg"] * 1e-3) / 1e12
        peak = flops / t_min / 1e12
        out = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
               "traffic": traffic, "ms_per_launch": d["ms_avg"], "launches": calls,
               "mix": {"fp32_mfma_flops": flops - f16 - f16x2, "f16x3_flops": f16, "f16x2_flops": f16x2,
                       "fp32_peak": FP32_MFMA_PEAK_TFLOPS, "f16_peak": F16_MFMA_PEAK_TFLOPS},
               # SURVEY 8(d)'s yardstick for the NN stage (all flops at the fp32 MFMA peak), for comparison only
               "frac_vs_fp32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS}
    else:
        ach = d["bytes"] / calls / (d["ms_avg"] * 1e-3) / 1e9
        out = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
               "traffic": traffic, "ms_per_launch": d["ms_avg"], "launches": calls}
        if flops > 0:
            out["matrix_side"] = {"tflops": flops / (d["ms_avg"] * 1e-3) / 1e12}
    # north_star: "rocprof HBM GB/s + MFMA utilisation reported against chip peak"
    out["hbm_gbs"] = (traffic / (d["ms_avg"] * 1e-3) / 1e9) if traffic else None    # PMC bytes per launch / HIP-event duration
    out["hbm_frac_of_peak"] = (out["hbm_gbs"] / HBM_PEAK_GBS) if traffic else None
    return out


def roofline_of(stages, traffic_by_stage, note=None, hbm_stages=()):
    """The dominant stage (by device time), priced by price_stage."""
    if not stages:
        return None
    dom = max(stages, key=lambda k: stages[k]["ms_total"])
    out = price_stage(dom, stages[dom], traffic_by_stage.get(dom), dom in hbm_stages)
    if note:
        out["note"] = note
    return out


def roofline_by_stage(stages, traffic_by_stage, util_of, min_share=0.10, hbm_stages=()):
    """Every stage that takes at least `min_share` of the step's device time, priced like `roofline` (so that the line does not depend on
    which of two stages wins by a tenth of a millisecond), each with its unit counters (`util_of(stage)` or None)."""
    if not stages:
        return None
    total = sum(v["ms_total"] for v in stages.values())
    out = {}
    for name in sorted(stages, key=lambda k: -stages[k]["ms_total"]):
        d = stages[name]
        if d["ms_total"] < min_share * total:
            continue
        ent = price_stage(name, d, traffic_by_stage.get(name), name in hbm_stages)
        ent["share_of_step"] = d["ms_total"] / total
        u = util_of(name) if util_of else None
        ent["unit_utilisation"] = u
        ent["mfma_util"] = (u or {}).get("MfmaUtil")
        out[name] = ent
    return out


def csrc_tree_hash():
    """`git rev-parse HEAD:sloika_amd/csrc` when the working tree's csrc/ is what HEAD has, else a hash of the files' contents -- what the
    committed counter files are stamped with (tools/stamp_profiles.py); None outside a git checkout without the files."""
    import hashlib
    import subprocess
    try:
        dirty = subprocess.run(["git", "status", "--porcelain", "--", "sloika_amd/csrc"], cwd=ROOT, stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, text=True, timeout=10)
        if dirty.returncode == 0 and not dirty.stdout.strip():
            r = subprocess.run(["git", "rev-parse", "HEAD:sloika_amd/csrc"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                               text=True, timeout=10)
            if r.returncode == 0 and r.stdout.strip():
                return "git:" + r.stdout.strip()
    except (OSError, subprocess.SubprocessError):
        pass
    return content_hash_of_csrc()


def content_hash_of_csrc():
    """sha256 over the names and contents of sloika_amd/csrc/* (the form that also works on the GPU box, which has no .git)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sloika_amd", "csrc")
    try:
        names = sorted(os.listdir(d))
    except OSError:
        return None
    for n in names:
        if n.endswith((".hip", ".h")):
            h.update(n.encode())
            with open(os.path.join(d, n), "rb") as fh:
                h.update(fh.read())
    return "sha256:" + h.hexdigest()[:16]


def counters_stale(table):
    """True when a committed counter file was taken on other kernels than the ones being run (its `csrc` stamp differs from the tree's);
    None when the file carries no stamp."""
    stamp = (table or {}).get("csrc")
    if not stamp:
        return None
    if isinstance(stamp, dict):                       # {"git": ..., "sha256": ...}: compare what this checkout can compute
        mine = content_hash_of_csrc()
        return None if mine is None else stamp.get("sha256") != mine.split(":", 1)[1]
    return stamp != content_hash_of_csrc()


def unit_utilisation(model, batch, chunk_len, kernel_substr, streams=1):
    """MfmaUtil / VALUBusy / LdsUtil / LDSBankConflict (percent of the kernel's duration, rocprofv3 derived counters, one pass per
    counter: tools/r04_measure.sh) of the kernel whose name contains `kernel_substr`, from the committed passes of this same workload;
    None when no file describes the configuration being run."""
    try:
        with open(os.path.join(ROOT, "profiles", "unit_utilisation.json")) as fh:
            table = json.load(fh)
    except (OSError, ValueError):
        return None
    for ent in table.get("workloads", []):
        if ent.get("workload") == [model, batch, chunk_len, streams]:
            for name, c in ent.get("kernels", {}).items():
                if kernel_substr in name:
                    # "stale": the passes were taken on other kernels than the ones running now (the file's csrc stamp against the tree's)
                    return {"kernel": name, "source": ent.get("source"), "stale": counters_stale(ent if ent.get("csrc") else table),
                            **{k: c[k] for k in ("MfmaUtil", "VALUBusy", "LdsUtil", "LDSBankConflict") if k in c}}
    return None


#: stages of the training step that stream their operands once (priced against HBM)
TRAIN_HBM_STAGES = ("train_wgrad", "train_dx", "train_xent")
#: the kernel a stage's time is spent in (for the utilisation lookup)
STAGE_KERNEL = {"gru_fused": "gru_bar16", "softmax_viterbi": "softmax_viterbi_kernel", "lstm_fused": "lstm_fused16_kernel",
                "gru_recurrent": "gru_scan", "gru_input_gemm": "gemm_rows_f16x3_kernel", "train_wgrad": "gemm_tn_bf16_multi", "train_gru_scan": "gru_bwd16_kernel",
                "train_dx": "gemm_bf16x6_kernel", "train_softmax_xent": "gemm_rows_f16x3_kernel"}


def pmc_traffic(model, batch, chunk_len):
    """HBM bytes per launch per stage from the committed rocprofv3 PMC passes of this same workload (tools/collect_pmc.sh);
    only quoted when a file describes the configuration being run."""
    for name in ("pmc_traffic.json", "pmc_traffic_%s_b%d.json" % (model, batch)):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                pmc = json.load(fh)
            if pmc.get("workload") == [model, batch, chunk_len]:
                PMC_STALE[(model, batch, chunk_len)] = counters_stale(pmc)
                return pmc.get("stage_bytes_per_launch", {})
        except (OSError, ValueError):
            pass
    return {}


#: (model, batch, chunk_len) -> whether the PMC traffic file quoted for it was taken on other kernels than the tree's (None: no stamp)
PMC_STALE = {}


def attach_counters(roof, model, batch, chunk_len, streams=1, tag=None):
    """Unit counters and the freshness of every looked-up figure for a priced stage (`roof` from price_stage)."""
    if roof is None:
        return None
    k = roof["kernel"]
    roof["unit_utilisation"] = unit_utilisation(tag or model, batch, chunk_len, STAGE_KERNEL.get(k, k), streams)
    roof["mfma_util"] = (roof["unit_utilisation"] or {}).get("MfmaUtil")
    roof["traffic_stale"] = PMC_STALE.get((model, batch, chunk_len)) if roof.get("traffic") else None
    return roof


class Runner(object):
    """One workload resident on the device: `nslot` Basecallers sharing a network, each with a stream, a few distinct input
    batches, pinned host buffers for the results.  step(i, nact) issues step i on stream i % nact."""

    def __init__(self, torch, model_name, B, L, nslot, rank=0, with_bases=False, main_stream=True):
        from sloika_amd import models, pipeline
        self.torch, self.B, self.L, self.with_bases = torch, B, L, with_bases
        self.net = models.randomise_zero_layers(models.build_model(model_name, klen=5, sd=0.5, seed=11))
        # borrow=True: every buffer of a call comes out of the Basecaller's own arena (the reference compiles its networks with
        # borrow=True, layers.py:34-36) -- a step allocates nothing, so the first steps run like the thousandth
        self.bcs = [pipeline.Basecaller(self.net, kmer_len=5, nbase=4, min_prob=1e-5, skip=0.0, borrow=True) for _ in range(nslot)]
        self.streams = ([torch.cuda.current_stream()] if main_stream else []) + \
            [torch.cuda.Stream() for _ in range(nslot - (1 if main_stream else 0))]
        self.nbuf = 2
        if self.net.insize == 1:
            self.host_in = [pipeline.synthetic_chunks(B, chunk_len=L, seed=0xdeadbeef, first_chunk=(rank * self.nbuf + i) * B)
                            for i in range(self.nbuf)]
        else:
            # event-feature models (baseline_gru / baseline_lstm / tiny_gru: a Window over 4 features per event): the input is the
            # [T, B, features] tensor calc_post takes; synthetic standard-normal features, L events per chunk
            self.host_in = [feature_input(L, B, self.net.insize, rank * self.nbuf + i) for i in range(self.nbuf)]
        self.dev = [torch.from_numpy(h).cuda() for h in self.host_in]
        first = self.net.layers[0]
        self.tout = first.out_len(L) if hasattr(first, "out_len") else L
        # two pinned result buffers per slot: the host reads the paths of call i - 2 of a slot (waits for that copy) before it issues
        # call i, whose copy then lands in the buffer just read
        self.out_host = [[torch.empty((B, self.tout), dtype=torch.int32).pin_memory() for _ in range(2)] for _ in range(nslot)]
        self.ncall = [0] * nslot
        self.copy_stream = torch.cuda.Stream()
        self.copied = [None] * nslot
        self.copied_before = [None] * nslot
        self.klen = 5
        if with_bases:
            self.bases_dev = [torch.empty((B, 5 * self.tout), dtype=torch.uint8, device="cuda") for _ in range(nslot)]
            self.nbases_dev = [torch.empty((B,), dtype=torch.int32, device="cuda") for _ in range(nslot)]
            self.bases_host = [torch.empty((B, 5 * self.tout), dtype=torch.uint8).pin_memory() for _ in range(nslot)]
            self.nbases_host = [torch.empty((B,), dtype=torch.int32).pin_memory() for _ in range(nslot)]
            self.acgt = int.from_bytes(b"ACGT".ljust(8, b"\0"), "little")
        # upload leg: the raw signal of the next batch comes from pinned host memory on the copy stream
        self.pinned_in = None
        self.upload_buf = None

    def set_deterministic(self, flag):
        """Basecaller(deterministic=...): True (the default) keeps every call on the plans that compute the same bits; False lets the
        sixteen-chunk Gru plan in where it is the faster one (four batches in flight, calls of more than 2048 chunks)."""
        for bc in self.bcs:
            bc.deterministic = bool(flag)

    def set_in_flight(self, n):
        for bc in self.bcs:
            bc.in_flight = n

    def step(self, i, nact=1, src=None):
        """The results leave for the host on a copy stream of their own: the next step's kernels do not queue behind a PCIe
        transfer.  The host is a consumer: before it issues call i of a slot it waits until the paths of call i - 2 of that slot
        are in pinned memory (where a caller reads them) -- which is also what makes reusing that call's result set (the
        Basecaller's arena keeps two, device.Arena) and its host buffer safe.  One whole step is always queued behind the running
        one, so the device never waits for the host; the host never runs more than two steps per slot ahead (measured,
        tools/warmup_curve.py: with the host ten steps ahead, every other of a fresh process's first twenty steps took 1 ms longer
        -- cross-stream waits on events that are far from firing -- and the first 25 steps averaged 3.95 ms against 3.70)."""
        torch = self.torch
        from sloika_amd import _lib, profiler
        k = i % nact
        with torch.cuda.stream(self.streams[k]):
            if self.copied_before[k] is not None:
                self.copied_before[k].synchronize()                   # the host has the paths of call i - 2 of this slot
            scores, paths, lens = self.bcs[k].call_chunks(self.dev[i % self.nbuf] if src is None else src)
            if self.with_bases:                                                 # base sequences, still on the device
                if self.copied[k] is not None:
                    self.streams[k].wait_event(self.copied[k])
                with profiler.region("bases", 0.0, 5.0 * paths.numel()):
                    _lib.check(_lib.lib().slk_paths_to_bases(paths.data_ptr(), paths.stride(0), lens.data_ptr(), self.B,
                                                             self.klen, 4, 1, self.acgt, self.bases_dev[k].data_ptr(),
                                                             self.klen * self.tout, self.nbases_dev[k].data_ptr(),
                                                             self.streams[k].cuda_stream), "paths_to_bases")
            done = torch.cuda.Event()
            done.record(self.streams[k])
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(done)
            self.out_host[k][self.ncall[k] & 1][:, : paths.shape[1]].copy_(paths, non_blocking=True)     # paths end up on the host
            self.ncall[k] += 1
            if self.with_bases:                                                      # ... and so do the base sequences
                self.bases_host[k].copy_(self.bases_dev[k], non_blocking=True)
                self.nbases_host[k].copy_(self.nbases_dev[k], non_blocking=True)
            self.copied_before[k] = self.copied[k]
            self.copied[k] = torch.cuda.Event()
            self.copied[k].record(self.copy_stream)

    def step_with_upload(self, i):
        """step i on buffer i % 2 while the raw signal of step i + 1 is uploaded from pinned host memory into the other buffer."""
        torch = self.torch
        if self.pinned_in is None:
            self.pinned_in = [torch.from_numpy(h).pin_memory() for h in self.host_in]
            self.upload_buf = [torch.empty_like(self.dev[0]) for _ in range(2)]
            self.uploaded = [None, None]
            self.consumed = [None, None]
            for j in range(2):
                self.upload_buf[j].copy_(self.pinned_in[j % self.nbuf], non_blocking=True)
        cur, nxt = i % 2, (i + 1) % 2
        main = self.streams[0]
        with torch.cuda.stream(self.copy_stream):
            if self.consumed[nxt] is not None:
                self.copy_stream.wait_event(self.consumed[nxt])               # the step that read this buffer has run
            self.upload_buf[nxt].copy_(self.pinned_in[(i + 1) % self.nbuf], non_blocking=True)
            self.uploaded[nxt] = torch.cuda.Event()
            self.uploaded[nxt].record(self.copy_stream)
        if self.uploaded[cur] is not None:
            main.wait_event(self.uploaded[cur])
        self.step(i, 1, src=self.upload_buf[cur])
        self.consumed[cur] = torch.cuda.Event()
        self.consumed[cur].record(main)


class StubRunner(object):
    """--stub-device: what Runner offers to the rank plumbing, with a step that sleeps (longer on higher ranks, so that the
    maximum over ranks and `per_rank_ms` have something to show)."""

    def __init__(self, rank):
        self.rank = rank

    def set_in_flight(self, n):
        pass

    def set_deterministic(self, flag):
        pass

    def step(self, i, nact=1, src=None):
        time.sleep(0.002 * (1.0 + 0.25 * self.rank))


class ClockProbe(object):
    """The shader clock under load: slk_clock_probe launched on a high-priority stream of its own between steps."""

    def __init__(self, torch, nmax=256):
        from sloika_amd import _lib
        self.torch, self.lib = torch, _lib.lib()
        self.stream = torch.cuda.Stream(priority=-1)
        self.buf = torch.zeros((nmax + 1, 2), dtype=torch.int64, device="cuda")
        self.n, self.nmax = 0, nmax
        # the stream's hardware queue is created at its first launch (milliseconds of host time during which the device runs dry, and a
        # few milliseconds without work cost the next steps: see main): first launch here, into a row of its own
        self.lib.slk_clock_probe(self.buf[nmax].data_ptr(), 1, self.stream.cuda_stream)
        self.stream.synchronize()

    def sample(self):
        if self.n < self.nmax:
            self.lib.slk_clock_probe(self.buf[self.n].data_ptr(), 64, self.stream.cuda_stream)
            self.n += 1

    def result(self):
        self.torch.cuda.synchronize()
        v = self.buf[: self.n].cpu().numpy().astype(np.float64)
        v = v[(v[:, 1] > 0)]
        if not len(v):
            return None
        mhz = v[:, 0] / v[:, 1] * 100.0
        return {"samples": int(len(mhz)), "min": float(mhz.min()), "mean": float(mhz.mean()), "max": float(mhz.max())}


def synthetic_reads(n, seed=0x5eed):
    """Whole reads with the length range of the reference's own test reads (test/unit/test_fast5.py:98-110: 51 129 ... 114 400
    samples): log-uniform lengths in [50 000, 115 000], signal cut from a pool of long synthetic traces."""
    from sloika_amd import pipeline
    rs = np.random.RandomState(seed)
    pool = pipeline.synthetic_chunks(48, chunk_len=120000, seed=seed)
    lens = np.exp(rs.uniform(np.log(50000.0), np.log(115000.0), size=n)).astype(np.int64)
    return [np.ascontiguousarray(pool[rs.randint(0, len(pool))][rs.randint(0, 5000):][:ln]) for ln in lens]


def leg_batch256(args, torch, B1=256, nfl=8):
    """The batch north_star quotes (256 chunks): BASELINE.json configs[1] and the metric's own model, one batch at a time and
    eight in flight, each with the roofline of its dominant kernel (HIP events, one batch at a time)."""
    from sloika_amd import profiler
    L = args.chunk_len

    def timed(fn, n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return time.perf_counter() - t

    small = {}
    for mname in (args.model,):                      # one model per child process (see the caller)
        r1 = Runner(torch, mname, B1, L, 1)
        ent = {"workload": "%s inference, %d-sample chunks, batch %d%s" % (
            mname, L, B1, " (BASELINE.json configs[1])" if mname == "baseline_raw_gru" and B1 == 256 else
            " (the architecture of the reference's only TRAINED model, models/pretrained.pkl: Conv 128 . Rev Gru 112 . Gru 144 . Rev Gru "
            "112; random weights)" if mname == "pretrained" else " (the metric's model at the north star's batch)")}
        r1.set_in_flight(1)
        # (a child process starts on a device that has been idle: ~50 launches pass before the device's power management lets the
        #  kernels run at their steady rate -- tools/warmup_kernel_only.py -- so the leg warms up for a dozen steps, and says so)
        nwarm = 12
        for i in range(nwarm):
            r1.step(i, 1)
        n = args.small_batch_steps
        d = timed(lambda i: r1.step(i, 1), n)
        ent["one_at_a_time"] = {"ms_per_step": d / n * 1e3, "value": B1 * L * n / d, "unit": "samples/s", "steps": n, "warmup": nwarm,
                                "streams_per_gpu": 1}
        rec1 = profiler.start()
        timed(lambda i: r1.step(i, 1), n)
        profiler.stop()
        st1 = rec1.summary()
        ent["stages_ms_per_step"] = {k: v["ms_total"] / n for k, v in sorted(st1.items())}
        tr1 = pmc_traffic(mname, B1, L)
        ent["roofline"] = attach_counters(roofline_of(st1, tr1, "one batch at a time: %d workgroups of 4 chunks on 256 CUs" % (B1 // 4)),
                                          mname, B1, L)
        if nfl <= 1:
            ent["roofline_by_stage"] = roofline_by_stage(st1, tr1, None)
            for v in (ent["roofline_by_stage"] or {}).values():
                attach_counters(v, mname, B1, L)
            small[mname] = ent
            del r1, rec1, st1
            continue
        ent["roofline_by_stage"] = roofline_by_stage(st1, tr1, None)
        for v in (ent["roofline_by_stage"] or {}).values():
            attach_counters(v, mname, B1, L)
        # ---- batches in flight, through the product's own API for a stream of batches (pipeline.Basecaller.call_batches: a fixed set of
        # streams, a Basecaller(borrow=True) per slot, the host a consumer): a short leg, then >= N seconds with the shader clock
        # sampled every 64 batches.  (Up to round 5 this leg drove eight Basecallers by hand; its 17 streams plus the API's 17 would be
        # more than the runtime's 32 hardware queues, and every queue of the process would be time-sliced.) ----
        from sloika_amd import pipeline
        r1.bcs = None                                     # (one at a time is done: its arena goes back)
        slots = pipeline.Basecaller.batch_slots(r1.net, nfl, kmer_len=5, nbase=4, min_prob=1e-5, skip=0.0)   # kept over the legs, as a server would
        kwb = dict(copy=False, slots=slots)

        def run_batches(nb=None, secs=None, probe=None):
            count = [0]
            t0 = time.perf_counter()

            def feed():
                while (count[0] < nb) if nb is not None else (time.perf_counter() - t0 < secs):
                    if probe is not None and count[0] % 64 == 63:
                        probe.sample()
                    count[0] += 1
                    yield r1.dev[count[0] % r1.nbuf]
            nres = sum(1 for _res in pipeline.Basecaller.call_batches(r1.net, feed(), **kwb))
            torch.cuda.synchronize()
            return nres, time.perf_counter() - t0
        run_batches(nb=2 * nfl)                          # first use: arenas, packs, streams
        n8, d = run_batches(nb=args.small_batch_steps * nfl)
        ent["eight_in_flight"] = {"ms_per_step": d / n8 * 1e3, "value": B1 * L * n8 / d, "unit": "samples/s", "steps": n8,
                                  "streams_per_gpu": nfl, "api": "pipeline.Basecaller.call_batches"}
        if args.sustained_seconds > 0:
            probe = ClockProbe(torch)
            nres, d = run_batches(secs=min(args.sustained_seconds, 10.0), probe=probe)
            ent["in_flight_sustained"] = {"seconds": d, "batches": nres, "ms_per_batch": d / nres * 1e3, "value": B1 * L * nres / d,
                                          "unit": "samples/s", "in_flight": nfl, "api": "pipeline.Basecaller.call_batches",
                                          "shader_clock_mhz": probe.result()}
        small[mname] = ent
        del r1, rec1, st1
    return small


def devices_of_ranks(torch, dist, bound, stub=False):
    """What every rank is bound to, in rank order: device name and PCI bus id (N ranks on N different GPUs show N different bus ids;
    the length of the list is the number of ranks the collective saw)."""
    if stub:
        mine = "stub:%d" % bound
    else:
        p = torch.cuda.get_device_properties(bound)
        bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
        mine = "%s@%s" % (p.name, bus)
    if dist is None:
        return [mine]
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, mine)
    return box


# ----------------------------------------------------------------------------------------------------------------------
def main_train(args, as_field=False, torch=None, dist=None):
    """One step = wrap_network's fg(x, labels, weights, rate) on one batch per GPU (bin/train_network.py:308).  With `as_field`
    it runs a few steps inside the inference line's process and returns the summary instead of printing a line."""
    from sloika_amd import _lib, models, profiler, shard, train
    if not as_field:
        import torch
        rank, world, local_rank = shard.dist_info()
        _lib.require_gpu()
        torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
        if "WORLD_SIZE" in os.environ:          # under a launcher (also with one rank: the RCCL path is the path that runs)
            import torch.distributed as dist
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        rank, world, _ = shard.dist_info()
    net = models.randomise_zero_layers(models.build_model(args.model, klen=5, sd=0.5, seed=11))     # same weights on every rank
    fg = train.wrap_network(net, min_prob=1e-30, l2=0.0, drop=20)                                  # train_network.py defaults
    B, L = args.batch, args.chunk_len
    rs = np.random.RandomState(1234 + rank)
    To = net.layers[0].out_len(L) if hasattr(net.layers[0], "out_len") else L        # event-feature models keep the length
    x = torch.from_numpy(rs.normal(size=(L, B, net.insize)).astype(np.float32)).cuda()
    labels = torch.from_numpy(rs.randint(0, net.size, size=(To, B)).astype(np.int32)).cuda()
    weights = torch.ones((To, B), dtype=torch.float32, device="cuda")
    steps = args.train_steps if as_field else args.steps
    warm = 5 if as_field else args.warmup         # (a leg behind others: the device's power management again needs a few steps)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warm):
        fg(x, labels, weights, 1e-3)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        loss, acc = fg(x, labels, weights, 1e-3 / (1.0 + i / 5000.0))
    barrier()
    dt = time.perf_counter() - t0
    allreduce_ms = None
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # the step's one exchange by itself: the all-reduce of a flat float32 gradient of this network's size (train.allreduce_mean_),
        # back to back, max over ranks
        flat = torch.zeros(sum(int(np.prod(p.get_value().shape)) for p in net.params()), dtype=torch.float32, device="cuda")
        for _ in range(3):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        barrier()
        ta = time.perf_counter()
        nar = 50
        for _ in range(nar):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        tar = torch.tensor([(time.perf_counter() - ta) / nar * 1e3], dtype=torch.float64, device="cuda")
        dist.all_reduce(tar, op=dist.ReduceOp.MAX)
        allreduce_ms = float(tar.item())
    devices = None if as_field else devices_of_ranks(torch, dist, torch.cuda.current_device())
    # per-stage HIP events in a pass of their own
    stages = {}
    if not args.no_stage_timing:
        rec = profiler.start()
        ns = max(2, min(steps, 5))
        for i in range(ns):
            fg(x, labels, weights, 1e-3)
        barrier()
        profiler.stop()
        stages = rec.summary()
        for v in stages.values():
            v["per_step"] = v["ms_total"] / ns
    roofline = None
    # HBM bytes per step of every stage from the committed PMC passes of this same workload (tools/collect_pmc.sh --train)
    stage_traffic, pmc = {}, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_train.json")) as fh:
            pmc = json.load(fh)
        if pmc.get("workload") == [args.model, args.batch, args.chunk_len]:
            stage_traffic = pmc.get("stage_bytes_per_step", {})
    except (OSError, ValueError):
        pass
    by_stage = None
    if stages:
        # the contractions of the backward pass (TRAIN_HBM_STAGES) stream their operands once and are bound by those loads (six bf16 MFMA
        # terms per product would take a quarter of the time at the matrix peak): priced against HBM on their ALGORITHMIC bytes; the scans
        # and the softmax layer on the matrix pipe at the number of MFMAs per product the kernels issue (train_gru_scan: two fp16 terms)
        per_launch = {k: stage_traffic[k] * ns / stages[k]["calls"] for k in stages if k in stage_traffic}
        tag = args.model + ":train"
        stale_tr = counters_stale(pmc) if stage_traffic else None
        roofline = roofline_of(stages, per_launch, hbm_stages=TRAIN_HBM_STAGES)
        by_stage = roofline_by_stage(stages, per_launch, None, hbm_stages=TRAIN_HBM_STAGES)
        for v in [roofline] + list((by_stage or {}).values()):
            attach_counters(v, args.model, B, L, tag=tag)
            v["traffic_stale"] = stale_tr if v.get("traffic") else None
            if v["bound"] == "hbm" and "matrix_side" in v:
                v["matrix_side"]["frac_of_bf16_peak_at_six_terms"] = 6.0 * v["matrix_side"]["tflops"] / F16_MFMA_PEAK_TFLOPS
    hbm_per_step = {k: stage_traffic[k] for k in sorted(stages) if k in stage_traffic} or None
    if as_field:
        return {"workload": "%s training step (forward, backward, ADAMski), %d-sample chunks, batch %d" % (args.model, L, B),
                "ms_per_step": dt / steps * 1e3, "value": world * B * L * steps / dt, "unit": "samples/s", "steps": steps,
                "final_loss": float(loss), "roofline": roofline, "roofline_by_stage": by_stage,
                "stages_ms_per_step": {k: v["per_step"] for k, v in sorted(stages.items())},
                "stages_hbm_bytes_per_step": hbm_per_step,
                "hbm_bytes_per_step": sum(hbm_per_step.values()) if hbm_per_step else None}
    if rank == 0:
        emit({
            "metric": "raw-signal samples/sec trained", "value": world * B * L * steps / dt, "unit": "samples/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (forward products and the weight-gradient-free GEMMs as 3-term fp16 splits, f32 accumulation)",
            "data": "synthetic",
            "config": {"workload": "%s training step (forward, backward, ADAMski), %d-sample chunks, batch %d per GPU, "
                                   "klen 5 (1025 states), drop 20" % (args.model, L, B),
                       "model": args.model, "chunk_len": L, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": "data parallel over %d GPU(s), one all-reduce of the flat gradient per step" % world},
            "roofline": roofline, "roofline_by_stage": by_stage,
            "cpu_baseline": cpu_baseline_train(args.model, L) if (world == 1 and args.cpu_chunks > 0) else None,
            "final_loss": float(loss),
            "devices": devices, "rccl_ranks_seen": len(devices) if dist is not None else None,
            "allreduce_ms_per_step": allreduce_ms,
            "gradient_floats": None if dist is None else int(flat.numel()),
            "stages_ms_per_step": {k: v["per_step"] for k, v in sorted(stages.items())},
            "stages_hbm_bytes_per_step": hbm_per_step,
            "hbm_bytes_per_step": sum(hbm_per_step.values()) if hbm_per_step else None})
    if dist is not None:
        dist.destroy_process_group()


#: what the driver keeps of this script's stdout is a few kilobytes of its tail: the LAST line is the record and stays under this
COMPACT_LINE_MAX = 4096


def _r(v, nd=5):
    """Floats of the compact line rounded to `nd` significant digits (22 KB of the round-5 line were 17-digit floats)."""
    if isinstance(v, float):
        return float("%.*g" % (nd, v))
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, nd) for x in v]
    return v


def compact_line(line):
    """The record: the contract's fields, the dominant stage's `roofline`, `cpu_baseline` and a dozen scalars of the other legs; every
    other figure of the run is in the detail file (`detail`)."""
    def g(d, *path):
        for p in path:
            if not isinstance(d, dict) or d.get(p) is None:
                return None
            d = d[p]
        return d
    roof = line.get("roofline")
    if roof:
        keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "ms_per_launch", "launches", "hbm_gbs", "mfma_util",
                "traffic_stale", "frac_vs_fp32_mfma_peak")
        u = roof.get("unit_utilisation") or {}
        roof = dict({k: roof.get(k) for k in keep}, stale=u.get("stale"),
                    units={k: u[k] for k in ("VALUBusy", "LdsUtil", "LDSBankConflict") if k in u} or None)
    cpu = line.get("cpu_baseline")
    if cpu:
        cpu = {"value": cpu.get("value"), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
               "sample": cpu.get("sample"), "single_core": g(cpu, "single_core", "value")}
    cfg = dict(line["config"])
    out = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    out["config"] = cfg
    out["roofline"] = roof
    out["cpu_baseline"] = cpu
    out["per_rank_ms"] = line.get("per_rank_ms")
    out["devices"] = line.get("devices")
    out["rccl_ranks_seen"] = line.get("rccl_ranks_seen")
    out["clock_mhz"] = g(line, "shader_clock_mhz_before_after", "mean")
    out["stages_ms"] = line.get("stages_ms_per_step")
    # one scalar per leg (samples/s unless the name says ms)
    out["sustained_one"] = g(line, "sustained", "one_at_a_time", "value")
    out["sustained_one_ms"] = g(line, "sustained", "one_at_a_time", "ms_per_step")
    out["four_in_flight_det"] = g(line, "sustained", "four_in_flight", "value")
    out["four_in_flight_nondet"] = g(line, "sustained", "four_in_flight_not_deterministic", "value")
    out["exact_f32"] = g(line, "exact_f32", "value")
    out["with_upload"] = g(line, "with_upload", "value")
    out["train_ms"] = g(line, "train", "ms_per_step")
    out["train_hbm_gb_per_step"] = (g(line, "train", "hbm_bytes_per_step") or line.get("hbm_bytes_per_step") or 0) / 1e9 or None
    out["pretrained"] = g(line, "pretrained", "one_at_a_time", "value")
    out["b256_baseline_raw_gru_one"] = g(line, "batch256", "baseline_raw_gru", "one_at_a_time", "value")
    out["b256_baseline_raw_gru_in_flight"] = g(line, "batch256", "baseline_raw_gru", "in_flight_sustained", "value") or \
        g(line, "batch256", "baseline_raw_gru", "eight_in_flight", "value")
    out["b256_rgrgr_one"] = g(line, "batch256", "raw_0.98_rgrgr", "one_at_a_time", "value")
    out["b256_rgrgr_in_flight"] = g(line, "batch256", "raw_0.98_rgrgr", "in_flight_sustained", "value") or \
        g(line, "batch256", "raw_0.98_rgrgr", "eight_in_flight", "value")
    out["whole_reads_resident"] = g(line, "whole_reads", "value")
    out["whole_reads_from_host"] = g(line, "whole_reads", "from_host_arrays", "value")
    out["allreduce_ms_per_step"] = line.get("allreduce_ms_per_step")
    out["csrc_tree"] = line.get("csrc_tree")
    out["detail"] = line.get("detail")
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline")
    out = _r({k: v for k, v in out.items() if v is not None or k in contract})
    s = json.dumps(out, separators=(",", ":"))
    if len(s) >= COMPACT_LINE_MAX:                      # never let the record outgrow what the driver keeps: shed the optional parts
        for k in ("stages_ms", "clock_mhz", "csrc_tree", "per_rank_ms"):
            out.pop(k, None)
            s = json.dumps(out, separators=(",", ":"))
            if len(s) < COMPACT_LINE_MAX:
                break
    return s


def emit(line):
    """Every figure of the run goes to `bench_detail.json` (beside this script, and into gpurun_out/ when that exists); stdout gets ONE
    line, the compact record, last."""
    paths = [os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = None
    for p in paths:
        try:
            with open(p, "w") as fh:
                json.dump(line, fh, indent=1)
            written = written or os.path.relpath(p, ROOT)
        except OSError:
            pass
    line["detail"] = written
    sys.stdout.flush()
    print(compact_line(line))
    sys.stdout.flush()


def children_first(args):
    """The legs that need a process of their own, run BEFORE this process has touched the device (no HIP call, no library load yet):
    the batch north_star quotes (256 chunks: BASELINE.json configs[1] and the metric's model, eight batches in flight through
    pipeline.Basecaller.call_batches) and the architecture of the reference's only trained model (models/pretrained.pkl: the widths 112 /
    144 that run projection GEMM + fp16-split scan).  A process of their own because eight batches in flight with the side streams of a
    birnn use 18 hardware queues; and FIRST because the queues of two processes add up: beside a parent that had used ~20 streams
    (in-flight legs, whole-read lanes, clock probes) baseline_raw_gru's eight batches in flight read 433-454 M samples/s, alone on the
    device 683-717 M (profiles/r06a_*, tools measured in one call).  -> {"batch256": ..., "pretrained": ...}"""
    import subprocess
    out = {"batch256": {}}
    me = os.path.abspath(__file__)
    for mname in ("baseline_raw_gru", "raw_0.98_rgrgr"):
        r = subprocess.run([sys.executable, me, "--only", "batch256", "--model", mname, "--small-batch-steps", str(args.small_batch_steps),
                            "--chunk-len", str(args.chunk_len), "--sustained-seconds", str(args.sustained_seconds)],
                           stdout=subprocess.PIPE, text=True)
        try:
            out["batch256"].update(json.loads(r.stdout.strip().split("\n")[-1]))
        except (ValueError, IndexError):
            out["batch256"][mname] = {"error": "child process failed (exit code %d)" % r.returncode}
    r = subprocess.run([sys.executable, me, "--only", "model1024", "--model", "pretrained", "--batch", "1024", "--small-batch-steps",
                        str(2 * args.small_batch_steps), "--chunk-len", str(args.chunk_len)], stdout=subprocess.PIPE, text=True)
    try:
        out["pretrained"] = json.loads(r.stdout.strip().split("\n")[-1])["pretrained"]
    except (ValueError, IndexError, KeyError):
        out["pretrained"] = {"error": "child process failed (exit code %d)" % r.returncode}
    return out


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
    if args.only == "batch256":
        from sloika_amd import _lib as _l
        _l.require_gpu()
        print(json.dumps(leg_batch256(args, torch)))
        return
    if args.only == "model1024":
        from sloika_amd import _lib as _l
        _l.require_gpu()
        print(json.dumps(leg_batch256(args, torch, B1=args.batch, nfl=1)))
        return
    from sloika_amd import shard
    rank, world, local_rank = shard.dist_info()
    stub = args.stub_device
    children = {}
    if (world == 1 and max(1, args.streams) == 1 and not args.quick and not stub and args.small_batch_steps > 0
            and args.model == "raw_0.98_rgrgr" and args.batch == 1024 and not args.with_bases):
        children = children_first(args)              # (nothing in this process has touched the device yet)
    if stub:
        # the rank plumbing alone (tests): gloo, host tensors, a Runner that sleeps; `device` records what a real run would bind
        ndev = int(os.environ.get("SLOIKA_AMD_STUB_DEVICES", "8"))
        bound = local_rank % max(1, ndev)
        dev = "cpu"
        sync = lambda: None
    else:
        from sloika_amd import _lib, layers as _layers, pipeline, profiler
        _lib.require_gpu()
        bound = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(bound)
        dev = "cuda"
        sync = torch.cuda.synchronize
    dist = None
    if "WORLD_SIZE" in os.environ:          # under a launcher (also with one rank: the RCCL path is the path that runs)
        import torch.distributed as dist
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    B, L = args.batch, args.chunk_len
    nstream = max(1, args.streams)
    extras = world == 1 and nstream == 1 and not args.quick and not stub        # the single-GPU legs behind the main region
    nslot = max(nstream, 4 if (extras and args.overlap_steps > 0) else 1)
    if stub:
        if os.environ.get("SLOIKA_AMD_STUB_FAIL_RANK") == str(rank):      # tests: a rank that dies must fail the whole launch
            sys.stderr.write("bench.py: rank %d fails on request\n" % rank)
            sys.exit(3)
        run = StubRunner(rank)
    else:
        run = Runner(torch, args.model, B, L, nslot, rank=rank, with_bases=args.with_bases, main_stream=(nstream == 1))
    run.set_in_flight(nstream)

    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    def reduce_max(dt):
        if dist is None:
            return dt
        tm = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        return float(tm.item())

    local_dt = [0.0]

    def timed(fn, n):
        barrier()
        t = time.perf_counter()
        for i in range(n):
            fn(i)
        sync()
        local_dt[0] = time.perf_counter() - t          # this rank's own work (per_rank_ms) ...
        barrier()
        return reduce_max(time.perf_counter() - t)     # ... and the job's: until the last rank is through

    # ---- the main region: W warm-up steps, then exactly K timed steps (no events inside) ----
    # (the device's power management: a few milliseconds WITHOUT work cost the next ~50 launches up to 12 % -- tools/warmup_kernel_only.py:
    # a pause of 5 ms behind 100 back-to-back launches of the Gru kernel, 572 ... 602 ... 525 us over the next 40 against 520 steady -- so
    # nothing sits between the warm-up steps and the timed region but the contract's barrier: the clock probe of the region's leading
    # edge is a launch on a stream of its own beside the last warm-up step, not a synchronise - probe - synchronise of its own)
    edge_probe = None if stub else ClockProbe(torch, nmax=4)
    for i in range(args.warmup):
        if edge_probe is not None and i == args.warmup - 1:
            edge_probe.sample()              # the shader clock right in front of the timed region ...
        run.step(i, nstream)
    if edge_probe is not None and args.warmup < 1:
        edge_probe.sample()
    dt = timed(lambda i: run.step(i, nstream), args.steps)
    if edge_probe is not None:
        edge_probe.sample()                  # ... and right behind it
    value = world * B * L * args.steps / dt
    # every rank's own time for the K steps (a straggler shows here; `ms_per_step` is the maximum)
    per_rank_ms = [local_dt[0] / args.steps * 1e3]
    if dist is not None:
        tl = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(tl, torch.tensor([per_rank_ms[0]], dtype=torch.float64, device=dev))
        per_rank_ms = [float(t.item()) for t in tl]

    devices = devices_of_ranks(torch, dist, bound, stub)
    # ---- the same steps once more with HIP events around every C-ABI call: per-stage times and the roofline ----
    stages, roofline, ms_profiled, by_stage = {}, None, None, None
    if not args.no_stage_timing and args.stage_steps > 0 and not stub:
        rec = profiler.start()
        dts = timed(lambda i: run.step(i, nstream), args.stage_steps)
        profiler.stop()
        stages = rec.summary()
        ms_profiled = dts / args.stage_steps * 1e3
        roofline = roofline_of(stages, pmc_traffic(args.model, B, L),
                               "latency-bound serial scan; %d chunks per CU, whose hi and lo state halves fill 8 of the 16 MFMA columns (twice over)" % max(1, B // 256)
                               if args.model == "raw_0.98_rgrgr" and B <= 1024 else None)
        by_stage = roofline_by_stage(stages, pmc_traffic(args.model, B, L), None)
        if roofline is not None:
            roofline["measured"] = "HIP events on the launch stream over %d steps issued right after the timed region " \
                                   "(the timed region itself carries no events)" % args.stage_steps
            attach_counters(roofline, args.model, B, L, nstream)
            for v in (by_stage or {}).values():
                attach_counters(v, args.model, B, L, nstream)

    def release():
        """torch's allocator caches device memory per stream; a leg that ran on side streams leaves tens of gigabytes reserved for
        streams that no longer exist, and the legs behind it then run into allocator garbage collection (measured: the training
        step 50 instead of 23 ms, the whole-read leg 330 instead of 510 M samples/s).  Every leg hands its cache back."""
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    line_extra = {}
    if extras:
        # ---- every product in plain float32 MFMA (no fp16 splits anywhere) ----
        if args.exact_steps > 0:
            keep = (_layers.SPLIT_F16, _layers.Softmax.split_f16)
            _layers.SPLIT_F16, _layers.Softmax.split_f16 = False, False
            try:
                run.step(0)
                dte = timed(lambda i: run.step(i), args.exact_steps)
                line_extra["exact_f32"] = {
                    "ms_per_step": dte / args.exact_steps * 1e3, "value": B * L * args.exact_steps / dte, "unit": "samples/s",
                    "steps": args.exact_steps,
                    "arithmetic": "float32 MFMA for every product (SLOIKA_AMD_EXACT_F32=1): two-kernel Gru, fp32 softmax "
                                  "projection + decoder on the logits; a correctness fallback, not a performance path"}
            finally:
                _layers.SPLIT_F16, _layers.Softmax.split_f16 = keep
            release()                                # (its 3.4 GB logits blocks would otherwise sit in the allocator's cache: see release)

        # ---- batches in flight ----
        if args.overlap_steps > 0:
            infl = {}
            for nact, key in ((2, "two_in_flight"), (4, "four_in_flight")):
                run.set_in_flight(nact)
                for i in range(2 * nact):
                    run.step(i, nact)
                n = args.overlap_steps * (nact // 2)
                d = timed(lambda i: run.step(i, nact), n)
                infl[key] = {"ms_per_step": d / n * 1e3, "value": B * L * n / d, "unit": "samples/s", "steps": n,
                             "streams_per_gpu": nact}
                if nact == 4:
                    # the price of "one set of bits per chunk whatever the plan" (Basecaller(deterministic=True), the default): the same leg
                    # with the sixteen-chunk Gru plan allowed (states equal to float32 rounding, up to 2 % of chunks called differently)
                    run.set_deterministic(False)
                    for i in range(2 * nact):
                        run.step(i, nact)
                    d = timed(lambda i: run.step(i, nact), n)
                    run.set_deterministic(True)
                    infl[key + "_not_deterministic"] = {"ms_per_step": d / n * 1e3, "value": B * L * n / d, "unit": "samples/s",
                                                        "steps": n, "streams_per_gpu": nact,
                                                        "note": "Basecaller(deterministic=False): sixteen chunks per Gru workgroup"}
            run.set_in_flight(1)
            if not args.with_bases:
                for mult, key in ((2, "two_as_one_batch"), (4, "four_as_one_batch")):
                    big = torch.cat([run.dev[i % run.nbuf] for i in range(mult)], dim=0)
                    outb = torch.empty((mult * B, run.tout), dtype=torch.int32).pin_memory()

                    copies = []

                    def step_big(_i, big=big, outb=outb, copies=copies):
                        if len(copies) >= 2:
                            torch.cuda.current_stream().wait_event(copies[-2])   # (the result set this call reuses: device.Arena)
                        scores, paths, lens = run.bcs[0].call_chunks(big)
                        done = torch.cuda.Event()
                        done.record(torch.cuda.current_stream())
                        with torch.cuda.stream(run.copy_stream):
                            run.copy_stream.wait_event(done)
                            outb[:, : paths.shape[1]].copy_(paths, non_blocking=True)
                            copies.append(torch.cuda.Event())
                            copies[-1].record(run.copy_stream)
                            del copies[:-2]
                    step_big(0)
                    n = max(1, args.overlap_steps // mult)
                    d = timed(step_big, n)
                    infl[key] = {"ms_per_step": d / (mult * n) * 1e3, "value": mult * B * L * n / d, "unit": "samples/s",
                                 "steps": mult * n, "chunks_per_call": mult * B,
                                 "note": "ms_per_step is per %d chunks; one call carries %d batches" % (B, mult)}
                    del big, outb
            line_extra["in_flight"] = infl
            release()

        # ---- the step including the upload of the next batch's raw signal (pinned host memory -> HBM on the copy stream) ----
        if args.upload_steps > 0 and not args.with_bases:
            for i in range(3):
                run.step_with_upload(i)
            d = timed(run.step_with_upload, args.upload_steps)
            line_extra["with_upload"] = {"ms_per_step": d / args.upload_steps * 1e3, "value": B * L * args.upload_steps / d,
                                         "unit": "samples/s", "steps": args.upload_steps,
                                         "upload_bytes_per_step": int(B * L * 4),
                                         "note": "float32 raw signal of the NEXT batch uploaded from pinned host memory on the "
                                                 "copy stream while this batch runs; `value` of the line excludes it"}

        # ---- what ONE host has to feed EIGHT ranks with: eight batches of raw signal per step (8 x B x L float32) from pinned memory
        # into HBM on the copy stream while the step runs.  (On an 8-GPU node every GPU has a link of its own; here all eight
        # uploads share this GPU's, so the leg bounds the host side -- pinned-memory reads, the copy engine's descriptors -- from above.)
        if args.host_feed_steps > 0 and not args.with_bases:
            nrank = 8
            pinned = [torch.from_numpy(run.host_in[j % run.nbuf]).pin_memory() for j in range(2)]
            land = [torch.empty_like(run.dev[0]) for _ in range(nrank)]
            feed_stream = torch.cuda.Stream()
            fed = [None]

            def step_fed(i):
                with torch.cuda.stream(feed_stream):
                    if fed[0] is not None:
                        feed_stream.wait_event(fed[0])
                    for r in range(nrank):
                        land[r].copy_(pinned[(i + r) % 2], non_blocking=True)
                run.step(i)
                fed[0] = torch.cuda.Event()
                fed[0].record(torch.cuda.current_stream())
            for i in range(2):
                step_fed(i)
            feed_stream.synchronize()
            t0 = time.perf_counter()
            sync()
            t0 = time.perf_counter()
            for i in range(args.host_feed_steps):
                step_fed(i)
            sync()
            feed_stream.synchronize()
            d = time.perf_counter() - t0
            nbytes = nrank * B * L * 4
            line_extra["host_feed"] = {"ranks_emulated": nrank, "bytes_per_step": int(nbytes), "steps": args.host_feed_steps,
                                       "ms_per_step": d / args.host_feed_steps * 1e3, "value": B * L * args.host_feed_steps / d,
                                       "unit": "samples/s (of the ONE batch that is computed)",
                                       "upload_gb_per_s": nbytes * args.host_feed_steps / d / 1e9,
                                       "demand_gb_per_s_at_value": nbytes / (dt / args.steps) / 1e9,
                                       "note": "eight ranks' uploads (pinned host memory -> HBM, copy stream) beside one rank's step; "
                                               "demand = what eight ranks at `value` each would ask of the host"}
            del pinned, land
            release()

        # (the batch-256 legs and the `pretrained` architecture ran in child processes BEFORE this process touched the device:
        #  children_first)
        line_extra.update(children)

        # ---- whole reads (the reference's inference mode), bucketed by length, ragged batches in flight ----
        if args.whole_reads > 0 and not args.with_bases:
            reads = synthetic_reads(args.whole_reads)
            kw = dict(kmer_len=5, skip=0.0)
            lanes = pipeline.Basecaller.read_lanes(run.net, 8, **kw)          # kept across calls, as a serving process would
            # two warm-up calls: torch's allocator caches device memory per stream, and the SECOND call of a process still asks the device
            # for memory (tools/whole_reads_alloc.py: 72 device allocations in the first call, 11 in the second -- the host runs ahead
            # differently once the first call's stalls are gone -- none from the third on; `allocator_during_the_call` says what the
            # timed call did).  A serving process is in that state after its first two read sets.
            for _ in range(2):
                pipeline.Basecaller.call_reads_bucketed(run.net, reads, max_batch=256, max_waste=0.08, lanes=lanes, **kw)
            torch.cuda.synchronize()
            # (1) from host arrays: trimming, bucketing, packing, upload, network, decoder, paths back on the host
            ms0 = torch.cuda.memory_stats()
            t0 = time.perf_counter()
            scores, paths, nsamp, wst = pipeline.Basecaller.call_reads_bucketed(run.net, reads, max_batch=256, max_waste=0.08,
                                                                                lanes=lanes, **kw)
            torch.cuda.synchronize()
            d_all = time.perf_counter() - t0
            ms1 = torch.cuda.memory_stats()
            alloc_delta = {k: int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries",
                                                                            "allocation.all.allocated")}
            alloc_delta["reserved_gb"] = round(ms1.get("reserved_bytes.all.current", 0) / 1e9, 1)
            # (2) the prepared batches resident in HBM (as the chunks of the main region are): network + decoder + paths to host
            batches, nsamp = pipeline.Basecaller.prepare_read_batches(run.net, reads, max_batch=256, max_waste=0.08, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            scores2, paths2 = pipeline.Basecaller.run_read_batches(run.net, batches, len(nsamp), lanes=lanes, **kw)
            torch.cuda.synchronize()
            d_dev = time.perf_counter() - t0
            assert all(np.array_equal(a, b) for a, b in zip(paths, paths2))
            line_extra["whole_reads"] = {
                "workload": "%d synthetic whole reads of 50k-115k samples (test/unit/test_fast5.py:98-110), per-read normalisation, "
                            "bucketed by length into ragged batches of <= 256 reads, all batches side by side on streams of their "
                            "own" % len(reads),
                "value": sum(nsamp) / d_dev, "unit": "samples/s", "seconds": d_dev, "reads_per_s": len(reads) / d_dev,
                "note": "`value`: padded batches resident in HBM when the clock starts (like the chunks of the main region), paths "
                        "on the host when it stops; `from_host_arrays`: the reads as numpy arrays on the host when the clock starts -- "
                        "bucketed by raw length, packed and uploaded bucket by bucket while the device runs the buckets before, "
                        "open-pore trimming on the device in the bucket's own stream (Basecaller._call_reads_streamed)",
                "from_host_arrays": {"value": sum(nsamp) / d_all, "unit": "samples/s", "seconds": d_all,
                                     "streamed": bool(wst.get("streamed")), "allocator_during_the_call": alloc_delta},
                "batches": wst["batches"], "padded_step_waste": wst["padded_step_waste"],
                "bases_called": int(sum(len(p) for p in paths))}
            del batches, paths2, lanes, scores, paths, scores2
            release()
            del reads

        # ---- a few steps of the training step (BASELINE.json configs[4] on this GPU) ----
        if args.train_steps > 0 and not args.with_bases:
            release()
            line_extra["train"] = main_train(args, as_field=True, torch=torch, dist=None)
            release()


        # ---- sustained load, LAST (the device throttles for a while after it: every other leg would pay): >= N seconds of
        # back-to-back steps, the shader clock sampled between steps ----
        if args.sustained_seconds > 0 and not args.with_bases:
            sus = {}
            for nact, key in ((1, "one_at_a_time"), (4, "four_in_flight"), (4, "four_in_flight_not_deterministic")):
                if nact > nslot:
                    continue
                run.set_deterministic(not key.endswith("not_deterministic"))     # (see in_flight: the sixteen-chunk Gru plan)
                run.set_in_flight(nact)
                for i in range(2 * nact):
                    run.step(i, nact)
                probe = ClockProbe(torch)
                barrier()
                t0 = time.perf_counter()
                n = 0
                while True:
                    for _ in range(16):
                        run.step(n, nact)
                        n += 1
                    probe.sample()
                    torch.cuda.current_stream().synchronize() if nact == 1 else run.streams[(n - 1) % nact].synchronize()
                    if time.perf_counter() - t0 >= args.sustained_seconds:
                        break
                barrier()
                d = time.perf_counter() - t0
                sus[key] = {"seconds": d, "steps": n, "ms_per_step": d / n * 1e3, "value": B * L * n / d, "unit": "samples/s",
                            "shader_clock_mhz": probe.result()}
            run.set_in_flight(1)
            run.set_deterministic(True)
            line_extra["sustained"] = sus


    if rank == 0:
        cpu = None
        if world == 1 and args.cpu_chunks > 0 and not stub:
            cpu = cpu_baseline(args.model, L, args.cpu_chunks)
        nst = max(1, args.stage_steps)
        gemm_flops = sum(stages[k]["flops"] for k in stages if k in MFMA_STAGES) / nst
        line = {
            "metric": "raw-signal samples/sec basecalled", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (products as 3-term fp16 splits with f32 accumulation, 22-bit operands; elementwise and DP in f32)",
            "data": "synthetic",
            "config": {"workload": "%s inference, %d-sample chunks, batch %d per GPU, klen 5 (1025 states), "
                                   "normalise->conv->GRU->softmax->Viterbi->paths%s on host; matrix products as fp16x3 splits "
                                   "(f32-grade, see exact_f32 for plain fp32 MFMA); softmax projection + decoder fused (logits "
                                   "never written)" % (args.model, L, B, " + base sequences" if args.with_bases else ""),
                       "model": args.model, "chunk_len": L, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": "chunks sharded over %d GPU(s), no collective" % world, "streams_per_gpu": nstream},
            "roofline": roofline,
            "roofline_by_stage": by_stage,
            "csrc_tree": csrc_tree_hash(),
            "per_rank_ms": per_rank_ms,
            "device_of_rank0": bound,
            "shader_clock_mhz_before_after": None if edge_probe is None else edge_probe.result(),
            "cpu_baseline": cpu,
            "stages_ms_per_step": {k: v["ms_total"] / nst for k, v in sorted(stages.items())},
            "ms_per_step_with_stage_events": ms_profiled,
            "e2e_algorithmic_tflops": (gemm_flops / (dt / args.steps) / 1e12) if stages else None,
        }
        line.update(line_extra)
        line["devices"] = devices
        line["rccl_ranks_seen"] = len(devices) if dist is not None else None
        if stub:
            line["data"] = "none: --stub-device (rank plumbing only, no GPU, no kernel)"
            line["value"] = None
        emit(line)
    elif stub:
        sys.stderr.write("bench.py: rank %d bound to device %d\n" % (rank, bound))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
