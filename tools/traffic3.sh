# HBM-side traffic of the config-3 chain (GPU box): FETCH_SIZE / WRITE_SIZE per kernel, separate --pmc passes.
#   bash tools/traffic3.sh [bits|bytes] [fast|mx]
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
LAY=${1:-bits}; FL=${2:-fast}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/tr3_$C
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/tr3_$C -o tr -- python3 tools/time_config3.py 1000 $LAY 3 $FL > gpurun_out/tr3_$C.log 2>&1
  echo "== $C ($LAY)"; python3 tools/pmc_summary.py gpurun_out/tr3_$C
done
