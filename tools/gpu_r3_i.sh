# round 3, call I: full GPU suite, scheduling bit-identity at full size, default bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 2700 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; tail -25 gpurun_out/pytest_gpu.log | cut -c1-200
timeout 1200 python tools/step_compare.py 512 8 8 > gpurun_out/r3_i_cmp512.log 2>&1
echo "step_compare rc=$?"; grep -v "^STEP" gpurun_out/r3_i_cmp512.log | cut -c1-160
SECONDS=0
timeout 900 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
echo "bench rc=$? elapsed $SECONDS s"
tail -1 gpurun_out/bench_default.log | cut -c1-5000
