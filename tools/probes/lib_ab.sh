# A/B of two builds of the library on one box: tools/probes/_old_lib.so against the in-tree one.
# usage: lib_ab.sh [pytest -k expression]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_nets_gpu.py tests/test_prod_shapes_gpu.py tests/test_blocks_gpu.py -m gpu -x -q -k "${1:-conv or macro or prod or block}" > gpurun_out/lib_ab_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/lib_ab_tests.log | cut -c1-200
cp se3ds_amd/csrc/libse3ds_hip.so /tmp/new_lib.so
for rep in 1 2; do
  for which in old new; do
    if [ $which == old ]; then cp tools/probes/_old_lib.so se3ds_amd/csrc/libse3ds_hip.so; else cp /tmp/new_lib.so se3ds_amd/csrc/libse3ds_hip.so; fi
    echo "== $which (rep $rep)"
    N=8 timeout 600 python tools/conv_bench.py 2>&1 | grep "TF/s" | cut -c1-170
  done
done
cp /tmp/new_lib.so se3ds_amd/csrc/libse3ds_hip.so
