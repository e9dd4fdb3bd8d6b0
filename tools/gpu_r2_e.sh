# round-2 fifth pass: single-pass binning (parity + A/B), fused perspective paths, cfg1 well-conditioned
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 1500 python -m pytest tests/test_warp_gpu.py \
  "tests/test_configs_gpu.py::test_cfg5_warp_1024x2048_two_views_bit_exact" \
  "tests/test_configs_gpu.py::test_autoregressive_rollout_vs_oracle" \
  "tests/test_configs_gpu.py::test_cfg1_generator_gradients_well_conditioned" \
  -m gpu -q -s --durations=8 -p no:cacheprovider > gpurun_out/r2_e.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"
grep -E "passed|failed|error" gpurun_out/r2_e.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r2_e.log | head -30
grep -aE "tensors|oracle gen|bf16 path|^  [a-z_]+/" gpurun_out/r2_e.log | head -40
grep -aE "^E  " gpurun_out/r2_e.log | head -30
for f in 1 0; do
  for d in random room; do
    SE3DS_SPLAT_FUSED=$f timeout 300 python bench.py --workload warp --warp-depth $d --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fused=$f depth=$d', 'ms/step %.4f' % d['ms_per_step'], 'proj us %.1f' % (1e3*r['ms_per_launch']), 'frac %.4f' % r['frac'])"
  done
done
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_warp_r2 -o warp -- python bench.py --workload warp --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/prof_warp_r2.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_warp_r2/warp_results.db gpurun_out/r02_warp_v0_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --workload warp --steps 200 --warmup 20 --no-cpu-baseline"
head -12 gpurun_out/r02_warp_v0_kernel_stats.csv | cut -c1-160
