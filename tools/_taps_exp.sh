cd $GRAFT_REPO_ROOT
SE3DS_WGRAD_TAPS3=0 python tools/wgrad_bench.py 2>/dev/null | tail -1
python tools/wgrad_bench.py 2>/dev/null | tail -1
python tools/wgrad_bench.py 512 512 3 32 64 8 2>/dev/null | tail -1
SE3DS_WGRAD_TAPS3=0 python tools/wgrad_bench.py 512 512 3 32 64 8 2>/dev/null | tail -1
python tools/wgrad_bench.py 128 128 3 256 512 8 2>/dev/null | tail -1
SE3DS_WGRAD_TAPS3=0 python tools/wgrad_bench.py 128 128 3 256 512 8 2>/dev/null | tail -1
timeout 600 python -m pytest tests/test_prod_shapes_gpu.py -q -k "bf16 and 3x3 and not partial" 2>&1 | tail -3
