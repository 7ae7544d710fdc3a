#!/bin/bash
# Counters of the sampler's kernels IN SITU: the CFG sampler of the bench (B=28, T=512, one hipGraph; 4 steps instead of 50 —
# a counter pass reads every dispatch back, 17 500 of them took more than 7 minutes) under rocprofv3,
# one run per counter group as MI355X_MICROARCH.md prescribes (--pmc with --kernel-trace only, the program itself after `--`).
# Run on the GPU box from the repo root:
#     bash tools/pmc_insitu.sh && python3 tools/pmc_insitu.py gpurun_out/pmc_insitu > profiles/r03/pmc_traffic.json
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/pmc_insitu
rm -rf $out && mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -- python3 tools/sampler_ab.py --steps 4 --warmup 0 --runs 1 > $out/$ctr.log 2>&1
  echo "pmc pass $ctr done"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out/SQ -- python3 tools/sampler_ab.py --steps 4 --warmup 0 --runs 1 > $out/SQ.log 2>&1
echo "pmc pass SQ done"
