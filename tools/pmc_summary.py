#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports exactly half of the bytes of
a wide coalesced streaming read (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE is exact for 16-B-per-lane
streaming stores.  Other access widths are uncalibrated, so the figures are upper-level estimates, not exact bytes.
"""
import collections
import csv
import glob
import json
import sys


def load(d):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return out


STAGES = {            # bench.py stage -> substrings of the kernels it launches (summed)
    "gru_fused": ("gru_bar16",),
    "softmax_gemm": ("gemm_rows_f16x3_kernel", "gemm_rows_kernel"),
    "viterbi": ("viterbi_forward", "viterbi_backtrace"),
    "softmax_viterbi": ("softmax_viterbi_kernel", "viterbi_backtrace"),
    "gemm_bias_act": ("gemm_bias_act_kernel", "gemm_rows_f16x3_kernel"),
    "conv1d": ("conv1d_",),
    "normalise": ("med_mad_",),
    # layers that run projection GEMM + scan (the `pretrained` architecture: 112 / 144 wide)
    "gru_input_gemm": ("gemm_rows_f16x3_kernel",),
    "gru_recurrent": ("gru_scan1t_kernel", "gru_scan16_kernel"),
}
#: stages whose launches of a step are DIFFERENT kernels (template instances): bytes per launch = launch-weighted mean over them
MEAN_STAGES = ("gru_input_gemm", "gru_recurrent")


def stage_file(summary_path, workload):
    """profiles/pmc_traffic.json (what bench.py quotes as roofline.traffic) from a per-kernel summary."""
    d = json.load(open(summary_path))
    out = {}
    for stage, keys in STAGES.items():
        ks = [v for k, v in d["kernels"].items() if any(s in k for s in keys)]
        tot = sum(v["hbm_bytes_per_launch"] for v in ks)
        if tot and stage in MEAN_STAGES:
            n = sum(v.get("launches_sampled", 1) for v in ks)
            tot = sum(v["hbm_bytes_per_launch"] * v.get("launches_sampled", 1) for v in ks) / max(1, n)
        if tot:
            out[stage] = tot
    res = {"workload": workload, "source": summary_path + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
           "tools/collect_pmc.sh; FETCH doubled per the gfx950 correction)", "stage_bytes_per_launch": out}
    if d.get("csrc"):
        res["csrc"] = d["csrc"]                      # tools/stamp_profiles.py: the kernels the passes were taken on
    json.dump(res, sys.stdout, indent=1)


TRAIN_STAGES = {      # bench.py --train stage -> substrings of the kernels it launches
    "gru_fused": ("gru_bar16",),
    "train_softmax_xent": ("gemm_rows_f16x3_kernel", "gemm_rows_kernel"),     # (round 4: both passes of slk_linear_xent_grad_f16x3)
    "conv1d": ("conv1d_",),
    "train_xent": ("softmax_xent_grad_kernel",),
    "train_xent_sums": ("reduce_sum_kernel", "reduce_rows_"),
    "train_gru_scan": ("gru_backward_", "gru_bwd16_kernel", "pack_x"),
    "train_wgrad": ("gemm_tn_", "tn_reduce_kernel", "im2col_"),
    "train_dx": ("gemm_bias_act_kernel", "act_backward_kernel", "gemm_bf16x6_kernel", "pack_bf16"),
}


def train_stage_file(summary_path, workload, steps_sampled=3):
    """profiles/pmc_traffic_train.json: HBM bytes per STEP of every training stage (launches per step x bytes per launch,
    summed over the stage's kernels; tools/collect_pmc.sh --train samples `steps_sampled` steps)."""
    d = json.load(open(summary_path))
    out, launches = {}, {}
    for stage, keys in TRAIN_STAGES.items():
        ks = [v for k, v in d["kernels"].items() if any(s in k for s in keys)]
        if ks:
            out[stage] = sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] / steps_sampled for v in ks)
            launches[stage] = sum(v["launches_sampled"] / steps_sampled for v in ks)
    res = {"workload": workload, "source": summary_path + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
           "tools/collect_pmc.sh --train; FETCH doubled per the gfx950 correction)", "stage_bytes_per_step": out,
           "kernel_launches_per_step": launches}
    if d.get("csrc"):
        res["csrc"] = d["csrc"]
    json.dump(res, sys.stdout, indent=1)


def main():
    if sys.argv[1] == "--stages":
        # python tools/pmc_summary.py --stages summary.json [model batch chunk_len]
        wl = [sys.argv[3], int(sys.argv[4]), int(sys.argv[5])] if len(sys.argv) >= 6 else ["raw_0.98_rgrgr", 1024, 4000]
        return stage_file(sys.argv[2], wl)
    if sys.argv[1] == "--train-stages":
        return train_stage_file(sys.argv[2], ["raw_0.98_rgrgr", 1024, 4000])
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [])))
        w = sum(write.get(k, [0])) / max(1, len(write.get(k, [])))
        res[k] = {"launches_sampled": len(fetch.get(k, [])), "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
                  "hbm_read_bytes": 2.0 * f * 1024.0, "hbm_write_bytes": w * 1024.0,
                  "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    json.dump({"note": "bytes per launch; FETCH_SIZE doubled per the gfx950 correction", "kernels": res}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
