# round 3, call N: resize branches; vectorised vs scalar wgrad reduce (A/B on one box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m pytest tests/test_warp_gpu.py -m gpu -x -q > gpurun_out/r3_n_tests.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"; tail -4 gpurun_out/r3_n_tests.log | cut -c1-200
for v in vec scalar vec scalar; do
  if [ $v = scalar ]; then export SE3DS_WGRAD_REDUCE_SCALAR=1; else unset SE3DS_WGRAD_REDUCE_SCALAR; fi
  timeout 600 python bench.py --no-cpu-baseline --no-warp --no-batch-max > gpurun_out/r3_n_bench_$v.log 2>&1
  echo "reduce=$v: $(tail -1 gpurun_out/r3_n_bench_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["by_kind"]["wgrad"]["tflops"], d["losses"]["gen/depth_loss"])')"
done
