cd $GRAFT_REPO_ROOT
for M in "" 1; do
MASK=$M python tools/wgrad_bench.py 512 512 3 32 64 8 2>/dev/null | tail -1
MASK=$M python tools/wgrad_bench.py 1024 1024 3 32 64 8 2>/dev/null | tail -1
MASK=$M python tools/wgrad_bench.py 256 256 3 64 128 8 2>/dev/null | tail -1
done
timeout 900 python -m pytest tests/test_prod_shapes_gpu.py -q -x -k "conv and bf16 and (partial or 3x3)" 2>&1 | tail -3
