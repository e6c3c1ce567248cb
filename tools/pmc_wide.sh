cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pw
  rocprofv3 --pmc $G --output-format csv -d /tmp/pw -- python3 $R/tools/time_wide.py $1 $2 > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
v=collections.defaultdict(list)
for f in glob.glob('/tmp/pw/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'rank_hist' in r['Kernel_Name'] or 'rank_pair' in r['Kernel_Name']: v[r['Counter_Name']].append(float(r['Counter_Value']))
for k,x in sorted(v.items()): print('%-24s %.6g  (per position %.1f)'%(k,sum(x)/len(x),sum(x)/len(x)/400000))
PY
done
