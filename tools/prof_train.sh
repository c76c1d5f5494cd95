#!/bin/bash
# usage: tools/prof_train.sh <tag> [bf16|off] [batch]   rocprofv3 kernel trace of a short train run (bench.py --mode train) and the
# per-kernel totals of its LAST step (tools/step_breakdown.py, marker = the matcher launch, once per step).
set -e
TAG=$1; AMP=${2:-bf16}; B=${3:-16}
O=$GRAFT_REPO_ROOT/gpurun_out/proftrain_$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --amp $AMP --batch $B --steps 3 --warmup 2 --cpu-images 0 > $O/bench.log 2>&1
T=$(ls $O/trace/*/*_kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/step_breakdown.py $T 70 matcher_cost > $O/step_breakdown.txt
cp $(ls $O/trace/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/trace   # the raw trace is tens of MB; the two summaries are what gets committed
tail -2 $O/bench.log
head -75 $O/step_breakdown.txt
