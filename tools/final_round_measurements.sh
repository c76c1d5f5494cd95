set -e
O=$GRAFT_REPO_ROOT/gpurun_out/final_r3
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
python bench.py --mode train --batch 8 --steps 3 --warmup 2 --cpu-images 0 --train-leg 0 > $O/bench_train_f32.json 2> $O/bench_train_f32.err
echo "train f32 done"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-images 0 --train-leg 0 > $O/prof_bench.log 2>&1
T=$(ls $O/trace/*/*_kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/step_breakdown.py $T 70 > $O/bench_step_breakdown.txt || true
cp $(ls $O/trace/*/*_kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
rm -rf $O/trace
echo "prof bench done"
cd $GRAFT_REPO_ROOT
tools/prof_train.sh r3final bf16 16 > $O/proftrain.log 2>&1
echo "prof train done"
