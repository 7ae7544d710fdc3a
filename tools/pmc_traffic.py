#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/pmc_traffic.sh into profiles/rNN/pmc_traffic.json.

    python3 tools/pmc_traffic.py gpurun_out/pmc > profiles/r02/pmc_traffic.json

Directory layout: <root>/<shape>_v<variant>_<COUNTER>/**/ *counter_collection.csv (one rocprofv3 run per counter: FETCH_SIZE
and WRITE_SIZE do not fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Per (shape, variant): mean
counter value over the dispatches of gemm_bf16_kernel in that run; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 — FETCH_SIZE
tallies 128-B requests at 64 B for 16-B-per-lane streams on gfx950 (same guide, "HBM").  Fabric-side bytes: Infinity-Cache
hits are included, so this bounds HBM traffic from above."""
import collections
import csv
import glob
import json
import os
import re
import sys

SHAPES = {"qkv": (1792, 1280), "out": (1280, 1280), "fc1": (5120, 1280), "fc2": (1280, 5120)}
EB = {"qkv": 2, "out": 4, "fc1": 2, "fc2": 4}     # output element bytes (bf16 / fp32 residual stream)
root = sys.argv[1]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 7168
acc = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(root, "*_v*_*"))):
    m = re.match(r"(\w+?)_v(\d+)_(\w+)$", os.path.basename(d))
    if not m:
        continue
    shape, variant, ctr = m.group(1), int(m.group(2)), m.group(3)
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm_bf16_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == ctr:
                vals.append(float(row["Counter_Value"]))
    if vals:
        acc[(shape, variant)][ctr] = (sum(vals) / len(vals), len(vals))
out = {"_how": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- python3 "
               "tools/gemm_bench.py --variants V --shapes S --rounds 1 --reps 2 (tools/pmc_traffic.sh); per-dispatch means; "
               "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 64 B per 128-B request for 16-B/lane "
               "streams: MI355X_MICROARCH.md 'HBM'); fabric-side (L2-miss) bytes, Infinity-Cache hits included"}
for (shape, variant), c in sorted(acc.items()):
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    N, K = SHAPES[shape]
    alg = M * K * 2 + N * K * 2 + M * N * EB[shape] * (2 if EB[shape] == 4 else 1)   # A + W + C (residual: read + write)
    out[f"{shape}_variant{variant}_M{M}_N{N}_K{K}"] = {
        "FETCH_SIZE_KB": round(c["FETCH_SIZE"][0], 1), "WRITE_SIZE_KB": round(c["WRITE_SIZE"][0], 1),
        "dispatches": c["FETCH_SIZE"][1], "traffic_bytes": int((2 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024),
        "algorithmic_bytes": alg}
print(json.dumps(out, indent=1))
