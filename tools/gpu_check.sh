# full GPU check: parity tests, smoke, short bench (run from the repo root on the GPU box)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/test.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/test.log
tail -5 gpurun_out/test.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout -k 10 600 python bench.py ${BENCH_ARGS:---steps 20 --warmup 3} > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/bench.log
tail -4 gpurun_out/bench.log
