#!/bin/bash
# SQ counter pass over the 512->512 @32x32 layer (B = 32): MFMA busy cycles, LDS activity, bank conflicts, busy cycles.
set -e
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/pmc_conv && mkdir -p gpurun_out/pmc_conv
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_conv -- python3 tools/lab/pmc_conv.py > gpurun_out/pmc_conv/out.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_conv/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "conv3x3_db" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, len(v), sum(v) / len(v))
PY
