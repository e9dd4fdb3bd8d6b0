# A/B of two builds of the library on one box: tools/probes/_old_lib.so (untracked) against the in-tree
# one, selected with SE3DS_LIB.   usage: lib_ab.sh [pytest -k expression]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
[ -f tools/probes/_old_lib.so ] || { echo "tools/probes/_old_lib.so missing"; exit 1; }
timeout 1500 python -m pytest tests/test_nets_gpu.py tests/test_prod_shapes_gpu.py tests/test_blocks_gpu.py -m gpu -x -q -k "${1:-conv or macro or prod or block}" > gpurun_out/lib_ab_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/lib_ab_tests.log | cut -c1-200
for rep in 1 2; do
  for which in old new; do
    if [ $which == old ]; then export SE3DS_LIB=$GRAFT_REPO_ROOT/tools/probes/_old_lib.so; else unset SE3DS_LIB; fi
    echo "== $which (rep $rep)"
    N=8 timeout 600 python tools/conv_bench.py 2>&1 | grep "TF/s" | cut -c1-170
  done
done
unset SE3DS_LIB
