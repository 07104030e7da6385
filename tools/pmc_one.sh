#!/bin/bash
# usage: tools/pmc_one.sh <tag> <launch> <op> <mode> <tile>   -> gpurun_out/pmc_<tag>/*.csv summaries
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/pmc_one.py "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 7):
    fs = glob.glob("$OUT/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs: print("no csv for pass", i); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        kn = row["Kernel_Name"]
        if "sg_" not in kn: continue
        agg[kn[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for kn, d in agg.items():
        print(kn, {c: round(sum(v[1:]) / max(len(v) - 1, 1)) for c, v in d.items()})
PY
