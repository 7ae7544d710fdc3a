#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) by kernel symbol + launch grid: calls, total ms, average us.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 tools/fwd_prof.py
    python tools/prof_agg.py gpurun_out/prof [--per N]      # --per: divide totals by N (e.g. forwards per run)
"""
import argparse, csv, glob, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("path")
ap.add_argument("--per", type=float, default=1.0)
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--skip-first", type=int, default=0, help="ignore the first N dispatches (set-up, warm-up)")
a = ap.parse_args()
files = [a.path] if a.path.endswith(".csv") else glob.glob(os.path.join(a.path, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    sys.exit(f"no *kernel_trace.csv under {a.path}")
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[a.skip_first:]
agg = {}
for r in rows:
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    key = (r["Kernel_Name"], grid)
    d = agg.setdefault(key, [0, 0.0])
    d[0] += 1
    d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot / 1e3:.3f} ms over {len(rows)} dispatches; per unit (/{a.per:g}): {tot / 1e3 / a.per:.3f} ms")
print(f"{'ms/unit':>9} {'calls/unit':>10} {'avg us':>9}  kernel")
for (name, grid), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"{us / 1e3 / a.per:9.3f} {n / a.per:10.1f} {us / n:9.1f}  {name[:150]}  [grid {grid}]")
