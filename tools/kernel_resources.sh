#!/bin/bash
# Registers, spills and LDS of every kernel in one .hip file:  tools/kernel_resources.sh sloika_amd/csrc/gru_fused16.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 | python3 -c '
import sys, re
cur = {}
for line in sys.stdin:
    m = re.search(r"remark: +(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur = {"name": t.split(":",1)[1].strip()[:70]}
    else:
        k, _, v = t.partition(":")
        if k.strip() in ("VGPRs", "VGPRs Spill", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"):
            cur[k.strip().split(" ")[0] if k.strip()!="VGPRs Spill" else "spill"] = v.strip()
if cur: print(cur)
'
