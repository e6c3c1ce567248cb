# per-launch PMC means of the K1 kernels of `bench.py $@` (1 M positions): bash tools/pmc_bench.sh [bench flags]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pw
  rocprofv3 --pmc $G --output-format csv -d /tmp/pw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --positions 1000000 "$@" > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
v=collections.defaultdict(list)
for f in glob.glob('/tmp/pw/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if any(k in n for k in ('ks_bucket','ks_rank','rank_hist','rank_pair')): v[(n.split('(')[0][-40:],r['Counter_Name'])].append(float(r['Counter_Value']))
for k,x in sorted(v.items()): print('%-42s %-22s %.6g  (per position %.1f)'%(k[0],k[1],sum(x)/len(x),sum(x)/len(x)/1000000))
PY
done
