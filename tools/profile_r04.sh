# Round-4 evidence (GPU box, repo root):  bash tools/profile_r04.sh [part ...]   parts: bench traffic sq stamps abl rate config3
# Kernel stats and counters always come from separate rocprofv3 runs (--pmc with --kernel-trace only).
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
PARTS=${@:-bench traffic sq stamps}
kstats() { # tag, program args...
  local tag=$1; shift
  rm -rf gpurun_out/ks_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -o $tag -- python3 "$@" > gpurun_out/ks_$tag.log 2>&1 || return 1
  cp $(find gpurun_out/ks_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/r04_${tag}_kernel_stats.csv
  tail -1 gpurun_out/ks_$tag.log | cut -c1-300
  head -4 gpurun_out/r04_${tag}_kernel_stats.csv | cut -c1-200
}
pmc() { # tag, counters, program args...
  local tag=$1 ctr=$2; shift 2
  rm -rf gpurun_out/pmc_$tag
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o $tag -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1 || return 1
  python3 tools/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/r04_pmc_$tag.txt && grep -A10 "bank_kernel" gpurun_out/r04_pmc_$tag.txt | head -12
}
for part in $PARTS; do
case $part in
bench)   # the default command's kernels under the profiler, headline mode and the int8 kernel beside it
  kstats bench_mx bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-modes --mode mx && \
  kstats bench_fast bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-modes --mode fast || exit 1 ;;
traffic) # HBM bytes per launch of the mx kernel on 64 corpus copies: 64 x 9.6 MB of bit rows = 614 MB >> 256 MiB of Infinity Cache
  pmc traffic_mx_fetch FETCH_SIZE bench.py --steps 3 --warmup 1 --replicas 64 --no-cpu-baseline --no-modes --mode mx && \
  pmc traffic_mx_write WRITE_SIZE bench.py --steps 3 --warmup 1 --replicas 64 --no-cpu-baseline --no-modes --mode mx && \
  python3 tools/traffic_json.py gpurun_out/r04_pmc_traffic_mx_fetch.txt gpurun_out/r04_pmc_traffic_mx_write.txt mx gpurun_out/r04_traffic_mx.json \
    "64 device-resident copies of the config-2 corpus; the mx kernel reads each copy's bit rows (9.6 MB) and group table (1.4 MB): 0.7 GB, 2.7x the 256 MiB Infinity Cache" \
    "--replicas 64 (tools/profile_r04.sh traffic)" || exit 1 ;;
sq)
  pmc sq1_mx "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" bench.py --steps 3 --warmup 1 --replicas 64 --no-cpu-baseline --no-modes --mode mx && \
  pmc sq2_mx "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" bench.py --steps 3 --warmup 1 --replicas 64 --no-cpu-baseline --no-modes --mode mx || exit 1 ;;
stamps)  # phase times of the mx kernel from the -DTSR_STAMP build (tools/build_stamp.sh)
  TSR_HIP_LIB=$PWD/template-speech-recognition_amd/lib/libtsr_stamp.so TSR_STAMP_OUT=gpurun_out/mx_stamps.txt timeout -k 10 200 python3 tools/time_fast.py 8 9 > /dev/null 2>&1 && \
  python3 tools/mx_stamp_report.py gpurun_out/mx_stamps.txt > gpurun_out/r04_mx_stamps.txt && cat gpurun_out/r04_mx_stamps.txt || exit 1 ;;
abl)     # development ablations (tools/abl_mx.sh build "1 4 5" first, here or locally)
  { bash tools/abl_mx.sh run "1 4 5"; python3 tools/time_fast.py 8 9 | tail -1; python3 tools/time_fast.py 8 1 | tail -1; } > gpurun_out/r04_mx_ablations.txt 2>&1; cat gpurun_out/r04_mx_ablations.txt ;;
config3) # BASELINE config 3 with the bank pass on the mx kernel (feature phases): stage times, kernel stats, HBM traffic of the chain
  { timeout -k 10 300 python3 tools/time_config3.py 1000 bits 5 fast | grep layout; timeout -k 10 300 python3 tools/time_config3.py 1000 bits 5 mx; } > gpurun_out/r04_config3_timing.txt 2>&1 && cat gpurun_out/r04_config3_timing.txt && \
  bash tools/kstat3.sh 1000 bits 5 mx > gpurun_out/r04_config3.log 2>&1 && cp gpurun_out/ks3/ks3_kernel_stats.csv gpurun_out/r04_config3_kernel_stats.csv && \
  bash tools/traffic3.sh bits mx > gpurun_out/r04_config3_traffic.txt 2>&1 && python3 tools/traffic3_json.py gpurun_out/r04_config3_traffic.txt gpurun_out/r04_config3_traffic.json mx || exit 1 ;;
rate)    # issue rate of the block-scaled matrix instruction
  timeout -k 10 120 ./tools/ubench/mx_rate.bin > gpurun_out/r04_mx_rate_ubench.txt 2>&1; cat gpurun_out/r04_mx_rate_ubench.txt ;;
esac
done
