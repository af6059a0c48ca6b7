cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | grep -E "^E|assert|Error|passed|failed" | head -30
