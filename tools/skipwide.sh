R=$GRAFT_REPO_ROOT
cp $R/nanomod_amd/libnanomod_hip.so /tmp/base.so
for S in 0 128 256 512 640; do
  if [ $S != 0 ]; then cp $R/nanomod_amd/libnanomod_hip_s$S.so $R/nanomod_amd/libnanomod_hip.so; fi
  TAG=skip$S timeout 120 python3 $R/tools/time_wide.py 50 800
  TAG=skip$S timeout 120 python3 $R/tools/time_wide.py 100 1000
done
cp /tmp/base.so $R/nanomod_amd/libnanomod_hip.so
