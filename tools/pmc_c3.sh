# SQ counters of the config-3 kernels (GPU box): bash tools/pmc_c3.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  tag=$(echo $pass | cut -c1-12 | tr ' ' '_')
  rm -rf gpurun_out/pmc_c3_$tag
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_c3_$tag -o e -- python3 tools/time_config3.py 1000 bits 2 > gpurun_out/pmc_c3_$tag.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_c3_$tag | grep -A9 "code_parts_mfma\|spread_mask_multi\|fast_bank_kernel"
done
