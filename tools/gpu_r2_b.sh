# round-2 second pass: re-run the fixed parity cases + new tests, then the default bench with shapes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 1500 python -m pytest tests/test_prod_shapes_gpu.py -k "stack3 or stack2 or encoder_final or conv1_7x7" \
  tests/test_configs_gpu.py tests/test_dist_gpu.py tests/test_warp_gpu.py \
  "tests/test_nets_gpu.py::test_adam_and_ema_recurrences_vs_oracle" -q -s --durations=12 -p no:cacheprovider \
  > gpurun_out/r2_b.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"
grep -E "passed|failed|error" gpurun_out/r2_b.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r2_b.log | head -30
grep -aE "^cfg|^R=|tensors|oracle" gpurun_out/r2_b.log | head -40
SECONDS=0
SE3DS_BENCH_SHAPES=1 timeout 900 python bench.py > gpurun_out/bench_shapes.log 2> gpurun_out/bench_shapes.err
echo "bench rc=$? elapsed $SECONDS s"
grep SHAPE gpurun_out/bench_shapes.log | head -60
tail -1 gpurun_out/bench_shapes.log | cut -c1-3500
