"""Per-kernel averages of the raw SQ counters collected by tools/r04_measure.sh (parts: sq) -> JSON on stdout.
Units (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_INSTS_* count wave instructions."""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
out = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(root, "sq_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not files:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k.split("(")[0][:90]][c] = round(sum(v) / len(v), 1)
        out[k.split("(")[0][:90]]["launches_sampled"] = len(v)
for k, c in out.items():
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if name in c:
                c[name + "_frac_of_wave_cycles"] = round(c[name] / wc, 4)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        c["lds_conflict_frac_of_lds_active"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
json.dump({"note": "rocprofv3 raw SQ counters per launch (averages over the sampled launches); one pass per counter set",
           "kernels": out}, sys.stdout, indent=1)
