cd $GRAFT_REPO_ROOT
for M in "" 1; do
MASK=$M python tools/wgrad_bench.py 2048 512 1 32 64 8 2>/dev/null | tail -1
MASK=$M python tools/wgrad_bench.py 512 2048 1 32 64 8 2>/dev/null | tail -1
MASK=$M python tools/wgrad_bench.py 512 512 3 32 64 8 2>/dev/null | tail -1
MASK=$M python tools/wgrad_bench.py 256 1024 1 64 128 8 2>/dev/null | tail -1
MASK=$M python tools/wgrad_bench.py 1024 256 1 64 128 8 2>/dev/null | tail -1
done
