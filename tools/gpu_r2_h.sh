# round-2 eighth pass: conditioned full-network tests, launcher tests, norm kernel micro-bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
for b in 8 16 32; do echo "== SE3DS_NORM_EW_BLOCKS=$b"; SE3DS_NORM_EW_BLOCKS=$b timeout 300 python tools/norm_kernels_bench.py 2>/dev/null; done
SECONDS=0
timeout 2400 python -m pytest \
  "tests/test_configs_gpu.py::test_autoregressive_rollout_vs_oracle" \
  "tests/test_configs_gpu.py::test_cfg2_inference_256x512_fp32_with_warp" \
  "tests/test_configs_gpu.py::test_cfg1_generator_gradients_well_conditioned" \
  "tests/test_configs_gpu.py::test_cfg1_lowres_train_g_d_fp32_and_bf16" \
  tests/test_dist_gpu.py \
  -m gpu -q -s --durations=8 -p no:cacheprovider > gpurun_out/r2_h.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"
grep -E "passed|failed|error" gpurun_out/r2_h.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r2_h.log | head -30
grep -aE "tensors|oracle gen|bf16 path|^cfg|^R=" gpurun_out/r2_h.log | head -40
grep -aE "^  [a-z_]+/" gpurun_out/r2_h.log | head -12
grep -aE "^E  " gpurun_out/r2_h.log | cut -c1-300 | head -30
