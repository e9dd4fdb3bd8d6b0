# round 3, call D: full GPU suite + warp A/B + kernel stats
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
for d in random room; do
  timeout 300 python bench.py --workload warp --warp-depth $d --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r3_warp_$d.log 2>&1
  echo "packed $d: $(tail -1 gpurun_out/r3_warp_$d.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"])')"
done
bash tools/gpu_prof_warp.sh random
cp gpurun_out/warp_kernel_stats.csv gpurun_out/r03_warp_kernel_stats.csv
SECONDS=0
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; tail -22 gpurun_out/pytest_gpu.log
