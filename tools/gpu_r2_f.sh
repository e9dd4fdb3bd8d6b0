# round-2 sixth pass: calibrated inference-state tests + warp A/B (single-pass 8 / 16 points, three-pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
for f in 8 16 0; do
  for d in random room; do
    SE3DS_SPLAT_FUSED=$f timeout 300 python bench.py --workload warp --warp-depth $d --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fused=$f depth=$d', 'ms/step %.4f' % d['ms_per_step'], 'proj us %.1f' % (1e3*r['ms_per_launch']), 'frac %.4f' % r['frac'])"
  done
done
SECONDS=0
timeout 1500 python -m pytest tests/test_warp_gpu.py -k "project or fused or banded" \
  "tests/test_configs_gpu.py::test_cfg5_warp_1024x2048_two_views_bit_exact" \
  "tests/test_configs_gpu.py::test_autoregressive_rollout_vs_oracle" \
  "tests/test_configs_gpu.py::test_cfg2_inference_256x512_fp32_with_warp" \
  "tests/test_configs_gpu.py::test_cfg1_generator_gradients_well_conditioned" \
  -m gpu -q -s --durations=8 -p no:cacheprovider > gpurun_out/r2_f.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"
grep -E "passed|failed|error" gpurun_out/r2_f.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r2_f.log | head -30
grep -aE "tensors:|oracle gen|bf16 path|^cfg2" gpurun_out/r2_f.log | head -40
grep -aE "^  [a-z_]+/" gpurun_out/r2_f.log | head -12
grep -aE "^E  " gpurun_out/r2_f.log | head -30
