# round-2 ninth pass: norm kernel A/B, cfg1 diagnostics (all tensors), well-conditioned test
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
for u in 1 2; do echo "== SE3DS_NORM_UNROLL=$u"; SE3DS_NORM_UNROLL=$u timeout 300 python tools/norm_kernels_bench.py 2>/dev/null; done
SECONDS=0
timeout 600 python -m pytest tests/test_nets_gpu.py -k "norm or generator or train or bf16_step" -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 2400 python -m pytest \
  "tests/test_configs_gpu.py::test_cfg1_generator_gradients_well_conditioned" \
  "tests/test_configs_gpu.py::test_cfg1_lowres_train_g_d_fp32_and_bf16" \
  -m gpu -q -s --durations=8 -p no:cacheprovider > gpurun_out/r2_i.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"
grep -E "passed|failed|error" gpurun_out/r2_i.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r2_i.log | head -30
grep -aE "tensors|oracle gen|bf16 path|^cfg|judged|BAD" gpurun_out/r2_i.log | cut -c1-200 | head -70
grep -aE "^E  " gpurun_out/r2_i.log | cut -c1-300 | head -12
