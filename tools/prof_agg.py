#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace results.db into per-kernel totals:  tools/prof_agg.py <results.db> [steps]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
# per-SHAPE rows: one kernel symbol serves several GEMM shapes; the launch grid tells them apart (grid = tiles x threads)
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
gcol = next((c for c in ("grid_size_x", "grid_size", "grid_x") if c in cols), None)
sel = f"s.kernel_name, d.start, d.end" + (f", d.{gcol}" if gcol else "")
rows = cur.execute(f"select {sel} from {kd} d join {ks} s on d.kernel_id=s.id").fetchall()
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n, a, b = r[0], r[1], r[2]
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if gcol:
        n = f"{n}  [grid {r[3]}]"
    agg[n][0] += 1
    agg[n][1] += b - a
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches; per step (/{steps:g}): {tot / 1e6 / steps:.3f} ms")
print(f"{'ms/step':>9} {'calls/step':>10} {'avg us':>9}  kernel")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:48]:
    print(f"{t / 1e6 / steps:9.3f} {c / steps:10.1f} {t / c / 1e3:9.1f}  {n[:150]}")
