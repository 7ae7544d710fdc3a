#!/usr/bin/env python3
"""Turn the in-situ rocprofv3 --pmc passes of tools/pmc_insitu.sh into profiles/rNN/pmc_traffic.json (+ an SQ table on stderr).

    python3 tools/pmc_insitu.py gpurun_out/pmc_insitu > profiles/r03/pmc_traffic.json 2> profiles/r03/pmc_sq_insitu.md

Per kernel symbol of the sampler's DiT block: mean counter value over all of its dispatches in the run (creation's eager step + the graph's 4 steps, x 28 layers).  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies a 128-B request of a 16-B-per-lane
stream at 64 B (MI355X_MICROARCH.md, "HBM").  Fabric-side bytes (L2 misses; Infinity-Cache hits included): an upper bound on
HBM traffic.  `_lib_sha256_16` names the libjat_hip.so the passes ran with; bench.py compares it with the library it loads."""
import collections, csv, glob, hashlib, json, os, re, sys

root = sys.argv[1]
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, D, MLP, NQ = 7168, 1280, 5120, 1792
SITES = [  # (key, regex on the kernel symbol, algorithmic bytes per launch)
    (f"fc1_variant38_M{M}_N{MLP}_K{D}", r"gemm_persist_kernel<2, 4, 7, 5, 2>", M * D * 2 + MLP * D * 2 + M * MLP * 2 + M * 16 * 4),
    (f"fc2_variant39_M{M}_N{D}_K{MLP}", r"gemm_kpair_kernel<3, 1", M * MLP * 2 + D * MLP * 2 + 2 * M * D * 4 + M * 16 * 4),
    (f"out_variant39_M{M}_N{D}_K{D}", r"gemm_kpair_kernel<3, 0", M * D * 2 + D * D * 2 + 2 * M * D * 4 + M * 16 * 4),
    (f"qkvattn_M{M}_N{NQ}_K{D}", r"gemm_bf16_kernel<2, 4, 4, 7, 8, 0, 6>", M * D * 2 + NQ * D * 2 + M * D * 2 + M * 16 * 4),
]


def rows(sub):
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        yield from csv.DictReader(open(f))


def mean_by_site(sub, ctr):
    acc = collections.defaultdict(list)
    for r in rows(sub):
        if r.get("Counter_Name") != ctr:
            continue
        for key, rx, _ in SITES:
            if re.search(rx, r.get("Kernel_Name", "")):
                acc[key].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = mean_by_site("FETCH_SIZE", "FETCH_SIZE"), mean_by_site("WRITE_SIZE", "WRITE_SIZE")
lib = os.path.join(HERE, "jatsr-just-audio-transformer-super-solution_amd", "csrc", "libjat_hip.so")
out = {"_how": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- python3 "
               "tools/sampler_ab.py --steps 4 --warmup 0 --runs 1 (tools/pmc_insitu.sh: the bench's CFG sampler, B=28 T=512, one hipGraph, 4 of its 50 steps); mean "
               "per dispatch of each kernel symbol; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 64 B per "
               "128-B request for 16-B/lane streams: MI355X_MICROARCH.md 'HBM'); fabric-side (L2-miss) bytes, Infinity-Cache hits included",
       "_lib_sha256_16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None}
for key, rx, alg in SITES:
    if key in fetch and key in write:
        out[key] = {"kernel": rx, "FETCH_SIZE_KB": round(fetch[key][0], 1), "WRITE_SIZE_KB": round(write[key][0], 1),
                    "dispatches": fetch[key][1], "traffic_bytes": int((2 * fetch[key][0] + write[key][0]) * 1024),
                    "algorithmic_bytes": alg}
print(json.dumps(out, indent=1))

# ---- SQ pass: MFMA-busy share and stall shares per kernel symbol, in situ --------------------------------------------------
cols = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT",
        "GRBM_GUI_ACTIVE"]
sq = {c: mean_by_site("SQ", c) for c in cols}
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "SQ", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for key, rx, _ in SITES:
            if re.search(rx, r.get("Kernel_Name", "")):
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
w = sys.stderr.write
w("| kernel (in the CFG sampler's graph, B=28 T=512) | launches | dur us (PMC run) | MFMA busy % of GRBM_GUI_ACTIVE x 1024 SIMDs / 8 XCDs | "
  "WAVE_CYCLES | WAIT_INST_ANY / WAVE_CYCLES | ACTIVE_INST_ANY / WAVE_CYCLES | LDS_BANK_CONFLICT |\n|---|---|---|---|---|---|---|---|\n")
for key, rx, _ in SITES:
    g = {c: sq[c].get(key, (float("nan"), 0))[0] for c in cols}
    n = sq[cols[0]].get(key, (0, 0))[1]
    d = sum(dur[key]) / len(dur[key]) if dur.get(key) else float("nan")
    busy = 100.0 * g["SQ_VALU_MFMA_BUSY_CYCLES"] / (g["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if g["GRBM_GUI_ACTIVE"] else float("nan")
    w(f"| {key} | {n} | {d:.1f} | {busy:.1f} | {g['SQ_WAVE_CYCLES']:.3g} | {g['SQ_WAIT_INST_ANY'] / g['SQ_WAVE_CYCLES']:.3f} | "
      f"{g['SQ_ACTIVE_INST_ANY'] / g['SQ_WAVE_CYCLES']:.3f} | {g['SQ_LDS_BANK_CONFLICT']:.3g} |\n")
