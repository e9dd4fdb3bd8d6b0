cd $GRAFT_REPO_ROOT
python tools/wgrad_bench.py 2>/dev/null | tail -1
