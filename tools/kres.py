#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: one line per kernel (VGPRs, SGPRs, scratch).
   hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip -o x.o 2> res.txt ; python tools/kres.py res.txt [filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = []
for ln in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]"):
        m = re.search(key + r": (\d+)", ln)
        if m and cur is not None:
            cur[key.split(" ")[0]] = int(m.group(1))
for r in rows:
    if flt in r["name"]:
        print(f"{r.get('VGPRs', -1):4d} v {r.get('TotalSGPRs', -1):4d} s {r.get('ScratchSize', -1):5d} scratch  {r['name']}")
