# round 3, call M: vectorised wgrad reduce; unsplit weight gradients under two streams (A/B)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 1200 python -m pytest tests/test_prod_shapes_gpu.py tests/test_golden_kernels.py -m gpu -x -q -k "wgrad or conv or golden" > gpurun_out/r3_m_tests.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; tail -4 gpurun_out/r3_m_tests.log | cut -c1-200
for sl in 256 128 256 128; do
  SE3DS_WGRAD_SLOTS=$sl timeout 600 python bench.py --no-cpu-baseline --no-warp --no-batch-max > gpurun_out/r3_m_bench_sl$sl.log 2>&1
  echo "wgrad_slots=$sl: $(tail -1 gpurun_out/r3_m_bench_sl$sl.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["by_kind"]["wgrad"]["tflops"], d["losses"]["gen/depth_loss"])')"
done
