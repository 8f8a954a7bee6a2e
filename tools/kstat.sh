cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
rm -rf gpurun_out/ks
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -o ks -- python3 tools/time_fast.py $1 > gpurun_out/ks.log 2>&1
tail -1 gpurun_out/ks.log
cat $(find gpurun_out/ks -name "*kernel_stats.csv" | head -1) | cut -c1-200
