cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
rm -rf gpurun_out/ks3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks3 -o ks3 -- python3 tools/time_config3.py ${1:-1000} ${2:-bits} ${3:-5} ${4:-fast} > gpurun_out/ks3.log 2>&1
tail -3 gpurun_out/ks3.log
cat $(find gpurun_out/ks3 -name "*kernel_stats.csv" | head -1) | cut -c1-220
