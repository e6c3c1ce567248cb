#!/bin/bash
# tools/build_variant.sh <name> "<extra -D flags>" [dtype=0] [all=1]: nanomod_amd/exp/<name>.so with one K1 instance file
# rebuilt under extra defines (the other objects come from the regular build)
set -e
cd "$(dirname "$0")/../nanomod_amd/csrc"
D=${3:-0}; A=${4:-1}
mkdir -p ../exp build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DNMOD_INST_DTYPE=$D -DNMOD_INST_ALL=$A $2 -c rank_stats_inst.hip -o /tmp/variant_$1.o
OBJS="build/nanomod_hip.o build/rank_order.o"
for d in 0 1; do for a in 0 1; do
  if [ $d = $D ] && [ $a = $A ]; then OBJS="$OBJS /tmp/variant_$1.o"; else OBJS="$OBJS build/rank_stats_d${d}_a${a}.o"; fi
done; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/$1.so $OBJS
echo built nanomod_amd/exp/$1.so
