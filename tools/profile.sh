# rocprofv3 kernel-trace + stats of the bench command (run on the GPU box from the repo root)
#   MODE=exact|fast  TAG=r01_exact  bash tools/profile.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
MODE=${MODE:-exact}; TAG=${TAG:-r01_$MODE}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --mode $MODE > gpurun_out/prof_$TAG.log 2>&1
echo "rocprof rc=$?"
tail -2 gpurun_out/prof_$TAG.log
find $OUT -name "*stats*.csv" | head
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/${TAG}_kernel_stats.csv && cat gpurun_out/${TAG}_kernel_stats.csv
