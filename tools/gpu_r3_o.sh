# round 3, call O: D pass-1 overlap: bit identity + A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
timeout 1500 python tools/step_compare.py 512 8 6 > gpurun_out/r3_o_cmp512.log 2>&1
echo "step_compare rc=$?"; grep -v "^STEP" gpurun_out/r3_o_cmp512.log | cut -c1-160 | tail -9
for ov in 1 0 1 0; do
  SE3DS_D_OVERLAP=$ov timeout 600 python bench.py --no-cpu-baseline --no-warp --no-batch-max > gpurun_out/r3_o_bench_ov$ov.log 2>&1
  echo "d_overlap=$ov: $(tail -1 gpurun_out/r3_o_bench_ov$ov.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["hbm_gib_peak"], d["losses"]["gen/depth_loss"])')"
done
SECONDS=0
timeout 1200 python -m pytest tests/test_nets_gpu.py tests/test_dist_gpu.py -m gpu -x -q --durations=6 > gpurun_out/r3_o_tests.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; tail -10 gpurun_out/r3_o_tests.log | cut -c1-200
