#!/bin/bash
# usage: tools/pmc.sh <tag> <kbench --only value> [extra kbench flags, e.g. --init] ; runs several rocprofv3 --pmc passes on tools/kbench.py
# (counters in their own runs: no tracing options beside --pmc)
set -e
TAG=$1; ONLY=$2; EXTRA=$3
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --only $ONLY --iters 3 $EXTRA > $O/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wm2f" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$O/summary.txt", "w") as out:
    for (k, c), v in sorted(agg.items()):
        line = f"{k:60s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
        print(line); out.write(line + "\n")
PY
