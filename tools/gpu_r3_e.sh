# round 3, call E: warp (hierarchical scans) + segment-optimizer A/B + full GPU suite
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_warp_gpu.py -m gpu -x -q > gpurun_out/r3_e_warp_tests.log 2>&1
echo "warp tests rc=$?"; tail -3 gpurun_out/r3_e_warp_tests.log
for rep in 1 2; do
for v in vec scalar; do
  if [ $v = scalar ]; then export SE3DS_PACK_SCALAR_OUT=1; else unset SE3DS_PACK_SCALAR_OUT; fi
  timeout 300 python bench.py --workload warp --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r3_warp_$v.log 2>&1
  echo "packed out=$v: $(tail -1 gpurun_out/r3_warp_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"])')"
done
done
unset SE3DS_PACK_SCALAR_OUT
timeout 300 python bench.py --workload warp --warp-depth room --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r3_warp_room.log 2>&1
echo "packed room: $(tail -1 gpurun_out/r3_warp_room.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"])')"
bash tools/gpu_prof_warp.sh random
cp gpurun_out/warp_kernel_stats.csv gpurun_out/r03_warp_kernel_stats.csv
for so in 1 0 1 0; do
  SE3DS_SEGMENT_OPTIMIZER=$so timeout 600 python bench.py --no-cpu-baseline --no-warp --no-batch-max > gpurun_out/r3_e_bench_so$so.log 2>&1
  echo "segment_opt=$so: $(tail -1 gpurun_out/r3_e_bench_so$so.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["hbm_gib_peak"], d["losses"])')"
done
SECONDS=0
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; tail -22 gpurun_out/pytest_gpu.log
