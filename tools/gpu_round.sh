# full verification pass: GPU tests, smoke, default bench (timed), forced-overlap bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; tail -25 gpurun_out/pytest_gpu.log
SECONDS=0
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$? elapsed $SECONDS s"; tail -3 gpurun_out/smoke.log
SECONDS=0
timeout 900 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
echo "bench rc=$? elapsed $SECONDS s"
tail -1 gpurun_out/bench_default.log | cut -c1-2500
SECONDS=0
SE3DS_FORCE_GRAD_SYNC=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_overlap.log 2>&1
echo "bench overlap rc=$? elapsed $SECONDS s"
tail -1 gpurun_out/bench_overlap.log | cut -c1-600
