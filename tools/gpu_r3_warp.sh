# round 3 warp: parity tests of the splat paths, A/B of packed vs 20-byte records, kernel stats
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 1200 python -m pytest tests/test_warp_gpu.py tests/test_blocks_gpu.py \
  "tests/test_configs_gpu.py::test_cfg5_warp_1024x2048_two_views_bit_exact" \
  -m gpu -x -q -s --durations=8 > gpurun_out/r3_warp_tests.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; grep -E "passed|failed|flips=|Error|error|assert" gpurun_out/r3_warp_tests.log | tail -40
for d in random room; do
  timeout 300 python bench.py --workload warp --warp-depth $d --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r3_warp_$d.log 2>&1
  echo "packed $d: $(tail -1 gpurun_out/r3_warp_$d.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"])')"
  SE3DS_SPLAT_PACKED=0 timeout 300 python bench.py --workload warp --warp-depth $d --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r3_warp_old_$d.log 2>&1
  echo "old    $d: $(tail -1 gpurun_out/r3_warp_old_$d.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"])')"
done
bash tools/gpu_prof_warp.sh random
cp gpurun_out/warp_kernel_stats.csv gpurun_out/r03_warp_kernel_stats.csv
bash tools/gpu_prof_warp.sh room
cp gpurun_out/warp_kernel_stats.csv gpurun_out/r03_warp_room_kernel_stats.csv
