# round-2 seventh pass: calibrated inference-state tests, warp tests, then the default bench + rocprof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 1800 python -m pytest tests/test_warp_gpu.py \
  "tests/test_configs_gpu.py::test_cfg5_warp_1024x2048_two_views_bit_exact" \
  "tests/test_configs_gpu.py::test_autoregressive_rollout_vs_oracle" \
  "tests/test_configs_gpu.py::test_cfg2_inference_256x512_fp32_with_warp" \
  "tests/test_configs_gpu.py::test_cfg1_generator_gradients_well_conditioned" \
  -m gpu -q -s --durations=8 -p no:cacheprovider > gpurun_out/r2_g.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"
grep -E "passed|failed|error" gpurun_out/r2_g.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r2_g.log | head -30
grep -aE "tensors:|oracle gen|bf16 path|^cfg2" gpurun_out/r2_g.log | head -40
grep -aE "^  [a-z_]+/" gpurun_out/r2_g.log | head -12
grep -aE "^E  " gpurun_out/r2_g.log | head -30
