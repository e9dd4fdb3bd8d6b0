cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -6 | cut -c1-400
