# round 3, call A: the new parity tests + the bench line with its new legs
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 1500 python -m pytest tests/test_blocks_gpu.py tests/test_dist_gpu.py \
  "tests/test_warp_gpu.py::test_fused_perspective_paths_vs_oracle_notebook_chain" \
  "tests/test_warp_gpu.py::test_fused_perspective_paths_match_the_op_chain" \
  "tests/test_configs_gpu.py::test_cfg5_warp_1024x2048_two_views_bit_exact" \
  "tests/test_configs_gpu.py::test_quantize_steps_bit_exact" \
  "tests/test_configs_gpu.py::test_cfg1_bf16_training_trajectory_tracks_fp32" \
  "tests/test_nets_gpu.py::test_grad_sync_drip_feeds_buckets" \
  "tests/test_nets_gpu.py::test_fused_spectral_fixup_clip_matches_separate_passes" \
  -m gpu -q -s --durations=15 > gpurun_out/r3_a_tests.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; grep -E "passed|failed|flips=|bf16|fp32|float32|Error|error" gpurun_out/r3_a_tests.log | tail -60
SECONDS=0
timeout 900 python bench.py > gpurun_out/r3_a_bench.log 2> gpurun_out/r3_a_bench.err
echo "bench rc=$? elapsed $SECONDS s"
tail -1 gpurun_out/r3_a_bench.log | cut -c1-6000
tail -5 gpurun_out/r3_a_bench.err
