cd $GRAFT_REPO_ROOT
SE3DS_BENCH_SHAPES=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep SHAPE | head -42
