cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -q -x --tb=line -k "conv" 2>&1 | tail -3 | cut -c1-300
python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids
