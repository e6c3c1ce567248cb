#!/bin/bash
# first GPU pass of round 3: the new bench presets + generator test (outputs under gpurun_out/)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -x -k "synth" > gpurun_out/r3a_pytest_synth.log 2>&1
for cfg in "ecoli:--steps 20 --warmup 5" "alltests:--config alltests --steps 20 --warmup 5 --no-cpu" \
           "i16:--dtype i16 --steps 20 --warmup 5 --no-cpu" "alli16:--config alltests --dtype i16 --steps 20 --warmup 5 --no-cpu" \
           "chr20:--config chr20 --steps 5 --warmup 2 --no-cpu" \
           "ragged:--config ragged --steps 3 --warmup 1 --no-cpu" "ragged_all:--config ragged --all-tests --steps 3 --warmup 1 --no-cpu" \
           "ragged_all_i16:--config ragged --all-tests --dtype i16 --steps 3 --warmup 1 --no-cpu" \
           "force:--force-collective --steps 10 --warmup 3 --no-cpu"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout 900 python bench.py $args > gpurun_out/r3a_bench_$name.json 2> gpurun_out/r3a_bench_$name.err
  echo "$name rc=$?"; tail -c 600 gpurun_out/r3a_bench_$name.err
done
