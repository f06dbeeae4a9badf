"""The per-workload utilisation summaries of tools/r04_measure.sh (gpurun_out/<tag>_<name>_unit_utilisation.json) as ONE table that
bench.py reads for `roofline.unit_utilisation` / `roofline.mfma_util`:   python tools/util_merge.py <tag> > profiles/unit_utilisation.json"""
import json
import os
import sys

tag = sys.argv[1]
root = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
WORKLOADS = {"b1024": ["raw_0.98_rgrgr", 1024, 4000, 1], "b1024x4": ["raw_0.98_rgrgr", 1024, 4000, 4],
             "b256_rgrgr": ["raw_0.98_rgrgr", 256, 4000, 1], "b256_baseline": ["baseline_raw_gru", 256, 4000, 1],
             "train": ["raw_0.98_rgrgr:train", 1024, 4000, 1], "pretrained": ["pretrained", 1024, 4000, 1]}
out = {"note": "rocprofv3 derived counters (MfmaUtil, VALUBusy, LdsUtil, LDSBankConflict: percent of the kernel's duration), one pass "
               "per counter with --kernel-trace only (tools/r05_measure.sh <tag> util); workload = [model, batch, chunk_len, streams]",
       "workloads": []}
for name, wl in WORKLOADS.items():
    path = os.path.join(root, "%s_%s_unit_utilisation.json" % (tag, name))
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    ent = {"workload": wl, "source": "profiles/%s_%s_unit_utilisation.json" % (tag, name), "kernels": d["kernels"]}
    if d.get("csrc"):
        ent["csrc"] = d["csrc"]                      # tools/stamp_profiles.py: the kernels the passes were taken on
    out["workloads"].append(ent)
json.dump(out, sys.stdout, indent=1)
