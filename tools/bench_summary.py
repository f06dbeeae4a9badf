import sys, json
d = json.loads(sys.stdin.read().strip().split(chr(10))[-1])
out = {"main": round(d["value"]/1e6)}
if "batch256" in d: out["b256"] = {m: (round(e["one_at_a_time"]["value"]/1e6), round(e["eight_in_flight"]["value"]/1e6)) for m, e in d["batch256"].items()}
if "whole_reads" in d: out["whole"] = (round(d["whole_reads"]["value"]/1e6), round(d["whole_reads"]["from_host_arrays"]["value"]/1e6))
if "train" in d: out["train_ms"] = round(d["train"]["ms_per_step"], 1)
if "sustained" in d: out["sustained"] = {k: round(v["value"]/1e6) for k, v in d["sustained"].items()}
if "in_flight" in d: out["in_flight"] = {k: round(v["value"]/1e6) for k, v in d["in_flight"].items()}
print(out)
