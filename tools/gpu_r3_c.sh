# round 3, call C: packed warp after the fence fix; dual-stream decoders A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m pytest tests/test_warp_gpu.py tests/test_blocks_gpu.py -m gpu -x -q --durations=5 > gpurun_out/r3_c_tests.log 2>&1
echo "pytest warp+blocks rc=$? elapsed $SECONDS s"; tail -4 gpurun_out/r3_c_tests.log
for d in random room; do
  timeout 300 python bench.py --workload warp --warp-depth $d --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r3_warp_$d.log 2>&1
  echo "packed $d: $(tail -1 gpurun_out/r3_warp_$d.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"])')"
done
bash tools/gpu_prof_warp.sh random
cp gpurun_out/warp_kernel_stats.csv gpurun_out/r03_warp_kernel_stats.csv
SECONDS=0
timeout 1500 python -m pytest tests/test_nets_gpu.py "tests/test_configs_gpu.py::test_cfg1_lowres_train_g_d_fp32_and_bf16" "tests/test_configs_gpu.py::test_cfg3_highres_512x1024_bf16_step" "tests/test_configs_gpu.py::test_cfg1_bf16_training_trajectory_tracks_fp32" -m gpu -x -q --durations=5 > gpurun_out/r3_c_nets.log 2>&1
echo "pytest nets (dual stream) rc=$? elapsed $SECONDS s"; tail -6 gpurun_out/r3_c_nets.log
for ds in 1 0 1 0; do
  SE3DS_DUAL_STREAM=$ds timeout 600 python bench.py --no-cpu-baseline --no-warp --no-batch-max > gpurun_out/r3_c_bench_ds$ds.log 2>&1
  echo "dual_stream=$ds: $(tail -1 gpurun_out/r3_c_bench_ds$ds.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["hbm_gib_peak"], d["losses"])')"
done
