"""Per-kernel averages of the derived utilisation counters collected by tools/collect_util.sh -> JSON on stdout."""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
out = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(root, "util_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not files:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k.split("(")[0][:90]][c] = round(sum(v) / len(v), 2)
        out[k.split("(")[0][:90]]["launches_sampled"] = len(v)
json.dump({"note": "rocprofv3 derived counters, percent of the kernel's duration; one pass per counter (tools/collect_util.sh)",
           "kernels": out}, sys.stdout, indent=1)
