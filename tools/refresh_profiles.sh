# Re-collect everything under profiles/ for the current build (GPU box, repo root).
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
MODE=fast TAG=r01_fast bash tools/profile.sh > gpurun_out/rp_fast.log 2>&1 && \
MODE=exact TAG=r01_exact bash tools/profile.sh > gpurun_out/rp_exact.log 2>&1 && \
MODE=fast24 TAG=r01_fast24 bash tools/profile.sh > gpurun_out/rp_fast24.log 2>&1 && \
MODE=fast bash tools/traffic.sh > gpurun_out/rp_traffic.log 2>&1 && \
bash tools/kstat3.sh 1000 > gpurun_out/rp_config3.log 2>&1 && \
cp gpurun_out/ks3/ks3_kernel_stats.csv gpurun_out/r01_config3_kernel_stats.csv && \
for m in fast fast24 exact; do timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --mode $m $([ $m = fast ] || echo --no-cpu-baseline) | tail -1 >> gpurun_out/r01_bench_all_modes.jsonl || exit 1; done
echo "refresh rc=$?"
