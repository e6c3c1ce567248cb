#!/bin/bash
# tools/sweep_narrow.sh: tools/sweep_narrow_threads.py for the build and for every library under nanomod_amd/exp/
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6m; echo "cpus $(nproc)  cgroup cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for LIB in $R/nanomod_amd/libnanomod_hip.so $(ls $R/nanomod_amd/exp/*.so 2>/dev/null); do
  NMOD_HIP_LIB=$LIB python3 $R/tools/sweep_narrow_threads.py "$@" 2>&1 | tee -a $R/gpurun_out/r6m/sweep_narrow.txt
done
