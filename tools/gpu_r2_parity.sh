# round-2 parity pass: production-shape convs, BASELINE configs, 2-replica oracle parity, Adam/EMA
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
free -g | head -2; nproc
SECONDS=0
timeout 2400 python -m pytest tests/test_prod_shapes_gpu.py tests/test_configs_gpu.py tests/test_dist_gpu.py \
  "tests/test_nets_gpu.py::test_adam_and_ema_recurrences_vs_oracle" -q -s --durations=25 -p no:cacheprovider \
  > gpurun_out/r2_parity.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"
grep -E "passed|failed|error" gpurun_out/r2_parity.log | tail -5
grep -E "^FAILED|^ERROR" gpurun_out/r2_parity.log | head -40
