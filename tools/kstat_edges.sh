cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
rm -rf gpurun_out/kedge
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kedge -o kedge -- python3 tools/time_edges.py 1000 --cpu > gpurun_out/kedge.log 2>&1
grep -v "^E2026\|^W2026" gpurun_out/kedge.log | tail -3
cat $(find gpurun_out/kedge -name "*kernel_stats.csv" | head -1) | cut -c1-70,100-200
