cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
python bench.py --workload warp --steps 20 --warmup 3 > gpurun_out/bench_warp.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_warp -o warp -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_warp.log 2>&1
cat gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3; tail -2 gpurun_out/bench_warp.log
find gpurun_out/prof_warp -name '*stats*' | head
