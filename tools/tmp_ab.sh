cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | grep -n "passed\|failed\|Error" | head -5
bash tools/ab.sh "--config chr20 --all-tests" nanomod_amd/exp/cur.so nanomod_amd/exp/p3.so
