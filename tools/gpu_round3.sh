# round 3, full verification pass: GPU suite (timed), smoke, default bench line, then the profiles
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; tail -24 gpurun_out/pytest_gpu.log | cut -c1-180
SECONDS=0
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$? elapsed $SECONDS s"; tail -3 gpurun_out/smoke.log
SECONDS=0
timeout 900 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
echo "bench rc=$? elapsed $SECONDS s"
tail -1 gpurun_out/bench_default.log | cut -c1-6000
bash tools/gpu_r3_prof.sh
