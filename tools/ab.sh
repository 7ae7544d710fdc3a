#!/bin/bash
# Same-box A/B of two builds of libjat_hip.so on the headline bench (box-to-box variance is +-4 %, so only numbers
# from ONE gpurun call compare):   bash tools/ab.sh <other libjat_hip.so> [rounds]
old=$1; n=${2:-2}
p='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"],1), round(d["forward"]["ms"],3))'
for i in $(seq $n); do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-long --no-train 2>/dev/null | python -c "$p" new
  JAT_LIB_PATH=$old python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-long --no-train 2>/dev/null | python -c "$p" old
done
