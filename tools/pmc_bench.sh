#!/bin/bash
# usage: tools/pmc_bench.sh <tag> ; HBM traffic of the K1 kernel INSIDE the model: rocprofv3 --pmc passes (counters only, one
# set per run, no tracing options) over bench.py, summarised per wm2f kernel.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
# FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -- the summary doubles it.
set -e
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out/pmcb_$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-images 0 --train-leg 0 > $O/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(list)
for f in glob.glob("$O/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "msdeform_stream" in k or "mask_einsum" in k or "masked_xattn_fwd" in k:
            agg[(re.sub(r"\(anonymous namespace\)::|void |wm2f::", "", k).split("(")[0][:56], r["Counter_Name"])].append(float(r["Counter_Value"]))
out = {}
for (k, c), v in sorted(agg.items()):
    out.setdefault(k, {})[c] = sum(v) / len(v)
    out[k]["launches_seen"] = len(v)
for k, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
json.dump(out, open("$O/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
