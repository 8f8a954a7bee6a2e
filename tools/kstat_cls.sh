cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
rm -rf gpurun_out/kcls
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kcls -o kcls -- python3 tools/time_classify.py 1000 > gpurun_out/kcls.log 2>&1
grep -v "^E2026\|^W2026" gpurun_out/kcls.log | tail -4
cat $(find gpurun_out/kcls -name "*kernel_stats.csv" | head -1) | cut -c1-70,160-260
