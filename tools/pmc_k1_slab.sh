#!/bin/bash
# usage: tools/pmc_k1_slab.sh <tag> "<case> <case> ..."   (cases as tools/k1_slab_inmodel.py: tm:0 hm:0 hm:800 ...)
# HBM traffic of the K1 launch inside the model per case: separate rocprofv3 --pmc passes (counters only), FETCH_SIZE doubled
# (gfx950: MI355X_MICROARCH.md, HBM).
set -e
TAG=$1
CASES=${2:-"hm:0 hm:800"}
O=$GRAFT_REPO_ROOT/gpurun_out/pmck1_$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in $CASES; do
  n=${c/:/_}
  for set in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    s=${set// /_}
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/${n}_$s -- python3 $GRAFT_REPO_ROOT/tools/k1_slab_inmodel.py --cases $c --rounds 1 --iters 2 > $O/${n}_$s.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob("$O/*/")):
    name = os.path.basename(d.rstrip("/"))
    case = "_".join(name.split("_")[:2])
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "msdeform_stream" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in agg.items():
            out.setdefault(case, {})[c] = sum(v) / len(v)
            out[case]["launches_seen"] = len(v)
for k, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
        d["x_algorithmic"] = round(d["hbm_bytes_per_launch"] / 550502400, 4)
json.dump(out, open("$O/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
