cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -q -x --tb=short -k "conv" 2>&1 | tail -15 | cut -c1-300
for m in ${MODES:-0 1}; do echo "== SE3DS_HALO_TILE=$m"; SE3DS_HALO_TILE=$m N=8 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids; done
