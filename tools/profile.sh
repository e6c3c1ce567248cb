#!/bin/bash
# Profiles bench.py's N=1 workload with rocprofv3 (run on the GPU box through gpurun).
# Kernel trace and every PMC group are separate runs, as the microarch guide prescribes.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_${1:-r1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/pmc_sq1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/pmc_sq2.err
find $OUT -name "*.csv" | head -40
# keep only small summaries: drop per-dispatch traces above 2 MB
find $OUT -name "*.csv" -size +2M -exec sh -c 'head -200 "$1" > "$1.head"; rm "$1"' _ {} \;
ls -la $OUT/*
