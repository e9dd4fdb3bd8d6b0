# round-2 fourth pass: the new row-f tests + the adjusted cfg1 tests
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 2000 python -m pytest tests/test_input_pipeline.py tests/test_warp_gpu.py \
  "tests/test_configs_gpu.py::test_autoregressive_rollout_vs_oracle" \
  "tests/test_configs_gpu.py::test_quantize_steps_bit_exact" \
  "tests/test_configs_gpu.py::test_cfg5_warp_1024x2048_two_views_bit_exact" \
  "tests/test_configs_gpu.py::test_cfg1_generator_gradients_well_conditioned" \
  "tests/test_configs_gpu.py::test_cfg1_lowres_train_g_d_fp32_and_bf16" \
  -m gpu -q -s --durations=8 -p no:cacheprovider > gpurun_out/r2_d.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"
grep -E "passed|failed|error" gpurun_out/r2_d.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r2_d.log | head -30
grep -aE "^cfg|^R=|tensors|oracle gen|^  [a-z_]+/" gpurun_out/r2_d.log | head -60
grep -aE "^E  " gpurun_out/r2_d.log | head -30
