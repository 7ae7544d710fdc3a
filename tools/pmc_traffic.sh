#!/bin/bash
# HBM-side traffic of the block's GEMM kernels, measured as MI355X_MICROARCH.md prescribes: one rocprofv3 run per counter
# (FETCH_SIZE / WRITE_SIZE), --kernel-trace only, the program itself after `--`.  Run on the GPU box from the repo root:
#     bash tools/pmc_traffic.sh && python3 tools/pmc_traffic.py gpurun_out/pmc > profiles/r02/pmc_traffic.json
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/pmc
mkdir -p $out
for spec in "fc1 31" "fc1 18" "fc2 32" "fc2 25" "out 32" "qkv 35"; do
  set -- $spec
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/${1}_v${2}_$ctr -- \
      python3 tools/gemm_bench.py --variants $2 --shapes $1 --rounds 1 --reps 2 > $out/${1}_v${2}_$ctr.log 2>&1
    echo "pmc pass $1 v$2 $ctr done"
  done
done
