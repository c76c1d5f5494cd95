#!/bin/bash
# usage: tools/pmc_k1_rows.sh <tag> [B] [bf16|f32] ; rocprofv3 --pmc passes (counters in their own runs) on tools/probes/k1_rows_bench.py:
# the training K1 op's forward / backward kernels at configs[2] size.  Summary -> gpurun_out/pmc_<tag>/summary.txt
set -e
TAG=$1; B=${2:-16}; DT=${3:-bf16}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $GRAFT_REPO_ROOT/tools/probes/k1_rows_bench.py $B $DT > $O/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wm2f" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-70:], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$O/summary.txt", "w") as out:
    for (k, c), v in sorted(agg.items()):
        line = f"{k:70s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
        print(line); out.write(line + "\n")
PY
rm -rf $O/p?/  # raw counter files: large; the summary is what gets committed
