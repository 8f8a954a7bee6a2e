# Round-3 evidence (GPU box, repo root):  bash tools/profile_r03.sh [part ...]   parts: bench traffic sq c45
# Kernel stats and counters always come from separate rocprofv3 runs (--pmc with --kernel-trace only).
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
PARTS=${@:-bench traffic sq c45}
kstats() { # tag, program args...
  local tag=$1; shift
  rm -rf gpurun_out/ks_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -o $tag -- python3 "$@" > gpurun_out/ks_$tag.log 2>&1 || return 1
  cp $(find gpurun_out/ks_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/r03_${tag}_kernel_stats.csv
  tail -2 gpurun_out/ks_$tag.log | cut -c1-400
  head -6 gpurun_out/r03_${tag}_kernel_stats.csv | cut -c1-200
}
pmc() { # tag, counters, program args...
  local tag=$1 ctr=$2; shift 2
  rm -rf gpurun_out/pmc_$tag
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o $tag -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1 || return 1
  python3 tools/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/r03_pmc_$tag.txt && cat gpurun_out/r03_pmc_$tag.txt
}
for part in $PARTS; do
case $part in
bench)
  kstats bench_fast bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-modes --mode fast && \
  kstats bench_fast24 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-modes --mode fast24 && \
  kstats bench_exact bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-modes --mode exact || exit 1 ;;
traffic)   # HBM bytes per launch on a working set of 16 corpus copies (1.2 GB >> 256 MiB of Infinity Cache)
  pmc traffic_fetch FETCH_SIZE bench.py --steps 3 --warmup 1 --replicas 16 --no-cpu-baseline --no-modes --mode fast && \
  pmc traffic_write WRITE_SIZE bench.py --steps 3 --warmup 1 --replicas 16 --no-cpu-baseline --no-modes --mode fast && \
  python3 tools/traffic_json.py gpurun_out/r03_pmc_traffic_fetch.txt gpurun_out/r03_pmc_traffic_write.txt fast gpurun_out/r03_traffic_fast.json || exit 1 ;;
sq)
  pmc sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" bench.py --steps 3 --warmup 1 --replicas 16 --no-cpu-baseline --no-modes --mode fast && \
  pmc sq2 "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" bench.py --steps 3 --warmup 1 --replicas 16 --no-cpu-baseline --no-modes --mode fast || exit 1 ;;
c45)
  M45=${M45:-fast,fast24,exact}
  timeout -k 10 300 python3 tools/time_config45.py $M45 4 5 > gpurun_out/r03_config45_timing.txt 2>&1 && cat gpurun_out/r03_config45_timing.txt && \
  kstats c45 tools/time_config45.py $M45 4 2 && \
  pmc c45_fetch FETCH_SIZE tools/time_config45.py $M45 4 1 && \
  pmc c45_write WRITE_SIZE tools/time_config45.py $M45 4 1 && \
  pmc c45_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" tools/time_config45.py $M45 4 1 || exit 1 ;;
c45x)  # the same bank with the round-robin block mapping (column groups of a segment on different XCDs)
  export TSR_XCD_MAP=0   # (read once, when the context is created)
  timeout -k 10 300 python3 tools/time_config45.py fast 4 5 > gpurun_out/r03_config45_roundrobin_timing.txt 2>&1 && cat gpurun_out/r03_config45_roundrobin_timing.txt && \
  pmc c45x_fetch FETCH_SIZE tools/time_config45.py fast 4 1 || exit 1
  unset TSR_XCD_MAP ;;
c3)   # BASELINE config 3: timing per layout, kernel stats and HBM-side traffic of the chain (bit form)
  timeout -k 10 300 python3 tools/time_config3.py 1000 bits,bytes 5 > gpurun_out/r03_config3_timing.txt 2>&1 && cat gpurun_out/r03_config3_timing.txt && \
  bash tools/kstat3.sh 1000 bits 5 > gpurun_out/r03_config3.log 2>&1 && cp gpurun_out/ks3/ks3_kernel_stats.csv gpurun_out/r03_config3_kernel_stats.csv && \
  bash tools/traffic3.sh bits > gpurun_out/r03_config3_traffic.txt 2>&1 && python3 tools/traffic3_json.py gpurun_out/r03_config3_traffic.txt gpurun_out/r03_config3_traffic.json || exit 1 ;;
misc)  # the rows either side of the path: classification, edges, mixture, the whole pipeline, per-call drop-ins
  true && \
  bash tools/kstat_cls.sh > gpurun_out/r03_classify.log 2>&1 && cp gpurun_out/kcls/kcls_kernel_stats.csv gpurun_out/r03_classify_kernel_stats.csv && \
  grep -v "^E2026\|^W2026" gpurun_out/kcls.log | tail -6 > gpurun_out/r03_classify_timing.txt && \
  timeout -k 10 300 python3 tools/time_mixture.py > gpurun_out/r03_mixture_timing.txt 2>&1 && \
  timeout -k 10 300 python3 tools/time_pipeline.py > gpurun_out/r03_pipeline_timing.txt 2>&1 && \
  bash tools/kstat_edges.sh > gpurun_out/r03_edges.log 2>&1 && cp gpurun_out/kedge/kedge_kernel_stats.csv gpurun_out/r03_edges_kernel_stats.csv && \
  grep -v "^E2026\|^W2026" gpurun_out/kedge.log | tail -3 > gpurun_out/r03_edges_timing.txt && \
  timeout -k 10 300 python3 tools/time_dropins.py > gpurun_out/r03_dropins_timing.txt 2>&1 || exit 1
  tail -3 gpurun_out/r03_config3.log; cat gpurun_out/r03_classify_timing.txt gpurun_out/r03_mixture_timing.txt gpurun_out/r03_pipeline_timing.txt gpurun_out/r03_edges_timing.txt gpurun_out/r03_dropins_timing.txt ;;
exactk)  # TSR_EXACT: the matrix-core kernel against the vector kernel, configs 2, 2b and 4
  { timeout -k 10 200 python3 tools/ab_kernel.py TSR_EXACTK valu,mfma 4 5 2 0 | tail -3 && \
    timeout -k 10 200 python3 tools/ab_kernel.py TSR_EXACTK valu,mfma 4 5 2b 0 | tail -3 && \
    timeout -k 10 200 python3 tools/ab_kernel.py TSR_EXACTK valu,mfma 2 3 4 0 | tail -3 && \
    timeout -k 10 200 python3 tools/ab_kernel.py TSR_FASTK old,tap2 8 7 2 | tail -3 && \
    timeout -k 10 200 python3 tools/ab_kernel.py TSR_FASTK old,tap2 4 5 4 | tail -3; } > gpurun_out/r03_kernel_ab.txt 2>&1 || exit 1
  cat gpurun_out/r03_kernel_ab.txt
  pmc exact_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-modes --mode exact || exit 1 ;;
esac
done
echo "profile_r03 done: $PARTS"
