# A/B of the 128-channel halo kernels: parity tests of the variants, then tools/conv_bench.py per variant
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_nets_gpu.py -m gpu -x -q -k "halo_128 or macro_tile" > gpurun_out/h4w_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/h4w_tests.log | cut -c1-200
for v in 0 3 4; do
  echo "== SE3DS_HALO_4W=$v"
  N=8 SE3DS_HALO_4W=$v timeout 600 python tools/conv_bench.py 2>&1 | grep "128->128" | cut -c1-200
done
