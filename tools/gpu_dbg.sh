cd $GRAFT_REPO_ROOT
python __graft_entry__.py smoke 2>&1 | tail -3
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -8 | cut -c1-400
