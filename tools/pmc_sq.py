#!/usr/bin/env python3
"""Summarise an SQ-counter rocprofv3 pass over tools/gemm_bench.py into a markdown table (per kernel symbol, mean per dispatch):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \\
        --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python3 tools/gemm_bench.py --variants 31,32 --shapes fc1,fc2 --rounds 1 --reps 2
    python3 tools/pmc_sq.py gpurun_out/pmc_sq
MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); SQ_WAVE_CYCLES & co. count quad-cycles
(MI355X_MICROARCH.md "rocprofv3 PMC slots" / cycle constants)."""
import collections, csv, glob, os, re, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "gemm_bf16_kernel" not in k:
            continue
        key = re.sub(r"\(.*", "", k).replace("void ", "") + f" grid {row.get('Grid_Size', '?')}"
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "gemm_bf16_kernel" in k:
            key = re.sub(r"\(.*", "", k).replace("void ", "") + f" grid {row.get('Grid_Size', '?')}"
            dur[key].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
cols = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"]
print("| kernel | dur us | clock GHz | MFMA busy % | " + " | ".join(c.replace("SQ_", "") for c in cols[1:6]) + " |")
print("|---|---|---|---|" + "---|" * 5)
for key, c in sorted(acc.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    d = sum(dur[key]) / len(dur[key]) if dur.get(key) else float("nan")
    cyc = m.get("GRBM_GUI_ACTIVE", float("nan")) / 8
    busy = 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / (cyc * 1024)
    wc = m.get("SQ_WAVE_CYCLES", float("nan"))
    def pct(x): return f"{m.get(x, float('nan')) / 1e6:.1f} M ({100 * m.get(x, float('nan')) / wc:.0f} %)"
    print(f"| {key[-70:]} | {d:.1f} | {cyc / d / 1e3:.2f} | {busy:.1f} | {wc / 1e6:.1f} M | {pct('SQ_WAIT_ANY')} | {pct('SQ_WAIT_INST_ANY')} | {pct('SQ_ACTIVE_INST_ANY')} | {m.get('SQ_LDS_BANK_CONFLICT', 0):.0f} |")
