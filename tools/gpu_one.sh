cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -k "$1" 2>&1 | tail -${2:-60}
