"""Stamp counter files with the kernels they were taken on: adds "csrc": "sha256:<hash of sloika_amd/csrc/*.hip, *.h>" to every JSON
file named (the form bench.py compares with the tree it runs on -- `"stale": true` in the bench line when they differ).  Run it where
the passes were collected, right after them (tools/r05_measure.sh does):      python tools/stamp_profiles.py gpurun_out/r05a_*.json
    --print     only print the tree's stamp"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    stamp = bench.content_hash_of_csrc()
    if "--print" in sys.argv:
        print(stamp)
        return
    for path in sys.argv[1:]:
        try:
            with open(path) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        if not isinstance(d, dict):
            continue
        d["csrc"] = stamp
        with open(path, "w") as fh:
            json.dump(d, fh, indent=1)
        print("stamped", path, stamp)


if __name__ == "__main__":
    main()
