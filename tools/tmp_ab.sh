cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest.log
bash tools/ab.sh "--config ragged --all-tests --dtype i16" nanomod_amd/exp/new.so nanomod_amd/exp/new2.so
bash tools/ab.sh "--config ragged --all-tests" nanomod_amd/exp/new.so nanomod_amd/exp/new2.so
