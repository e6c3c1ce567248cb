#!/bin/bash
# PMC passes over bench.py's N=1 workload (separate runs per counter group); prints per-launch means for K1.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcall_${1:-x}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { rocprofv3 --pmc "$@" --output-format csv -d $OUT/$1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --all-tests > /dev/null 2> $OUT/$1.err; }
run SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
run SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob('$OUT/*/*/*counter_collection.csv')):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'rank_stats' in r['Kernel_Name'] or 'ks_rank' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value'])); meta=(r['Kernel_Name'][:60],r['Grid_Size'],r['LDS_Block_Size'],r['VGPR_Count'])
    print(meta)
    for k,v in sorted(d.items()): print('  %-24s %.6g'%(k,sum(v)/len(v)))
PY
