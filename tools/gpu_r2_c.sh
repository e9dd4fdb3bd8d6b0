# round-2 third pass: config-level tests, 2-replica parity, warp tests, Adam/EMA recurrences
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 2000 python -m pytest tests/test_configs_gpu.py tests/test_dist_gpu.py tests/test_warp_gpu.py \
  "tests/test_nets_gpu.py::test_adam_and_ema_recurrences_vs_oracle" -q -s --durations=12 -p no:cacheprovider \
  > gpurun_out/r2_c.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"
grep -E "passed|failed|error" gpurun_out/r2_c.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r2_c.log | head -30
grep -aE "^cfg|^R=|tensors|oracle" gpurun_out/r2_c.log | head -40
grep -aE "^E  " gpurun_out/r2_c.log | head -40
