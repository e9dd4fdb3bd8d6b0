# round 3, call J: wgrad-stream bit identity + A/B, the tests touched since call I
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
timeout 1500 python tools/step_compare.py 512 8 6 > gpurun_out/r3_j_cmp512.log 2>&1
echo "step_compare rc=$?"; grep -v "^STEP" gpurun_out/r3_j_cmp512.log | cut -c1-160
for wg in 1 0 1 0; do
  SE3DS_WGRAD_STREAM=$wg timeout 600 python bench.py --no-cpu-baseline --no-warp --no-batch-max > gpurun_out/r3_j_bench_wg$wg.log 2>&1
  echo "wgrad_stream=$wg: $(tail -1 gpurun_out/r3_j_bench_wg$wg.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["hbm_gib_peak"], d["losses"])')"
done
SECONDS=0
timeout 1500 python -m pytest tests/test_warp_gpu.py tests/test_dist_gpu.py "tests/test_nets_gpu.py::test_adam_and_ema_recurrences_vs_oracle" "tests/test_nets_gpu.py::test_scheduling_switches_are_bit_identical" "tests/test_nets_gpu.py::test_segment_grad_sync_matches_serial_path" "tests/test_configs_gpu.py::test_cfg1_generator_gradients_vs_fp64_yardstick" -m gpu -x -q --durations=8 > gpurun_out/r3_j_tests.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; tail -16 gpurun_out/r3_j_tests.log | cut -c1-200
