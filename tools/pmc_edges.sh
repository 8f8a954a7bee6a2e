# SQ counters of the edge kernel (GPU box): bash tools/pmc_edges.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_ANY"; do
  tag=$(echo $pass | cut -c1-12 | tr ' ' '_')
  rm -rf gpurun_out/pmc_edge_$tag
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_edge_$tag -o e -- python3 tools/time_edges.py 1000 > gpurun_out/pmc_edge_$tag.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_edge_$tag | grep -A12 "edge_block"
done
