cd $GRAFT_REPO_ROOT
bash tools/ab.sh "--config ragged" nanomod_amd/exp/pipe3.so nanomod_amd/exp/pipe4.so
