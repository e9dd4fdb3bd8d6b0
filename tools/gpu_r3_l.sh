# round 3, call L: coarser segments + batched operand refresh: bit identity, A/B, kernel stats
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m pytest "tests/test_nets_gpu.py::test_segment_grad_sync_matches_serial_path" "tests/test_nets_gpu.py::test_scheduling_switches_are_bit_identical" "tests/test_nets_gpu.py::test_train_g_d_gradients_and_update_fp32" tests/test_dist_gpu.py -m gpu -x -q --durations=5 > gpurun_out/r3_l_tests.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; tail -6 gpurun_out/r3_l_tests.log | cut -c1-200
for so in 1 0 1 0; do
  SE3DS_SEGMENT_OPTIMIZER=$so timeout 600 python bench.py --no-cpu-baseline --no-warp --no-batch-max > gpurun_out/r3_l_bench_so$so.log 2>&1
  echo "segment_opt=$so: $(tail -1 gpurun_out/r3_l_bench_so$so.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["hbm_gib_peak"], d["losses"])')"
done
rm -rf gpurun_out/prof_gan_r3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gan_r3 -o gan -- python bench.py --no-cpu-baseline --no-batch-max --no-warp > gpurun_out/prof_gan_r3.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_gan_r3/gan_results.db gpurun_out/r03_gan_step_b8_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-batch-max --no-warp   (3 warm-up + 10 timed + 1 instrumented train_g_d step = 14 steps; model build kernels included; decoders on two streams, per-module optimiser on a side stream)"
tail -1 gpurun_out/prof_gan_r3.log | cut -c1-200
head -36 gpurun_out/r03_gan_step_b8_kernel_stats.csv | cut -c1-130
rm -rf gpurun_out/prof_gan_r3
