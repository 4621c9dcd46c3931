cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r05_clock
mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/tools/run_iqft_exact.py > $OUT/log.txt 2> $OUT/err.txt
echo rc=$?
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_r05_clock/pmc/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(dict)
for r in rows:
    if 'x8' in r['Kernel_Name']:
        by[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
        by[r['Dispatch_Id']]['_t'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) if 'End_Timestamp' in r else 0
for d, c in list(by.items())[-3:]:
    print(d, {k: f"{v:.4g}" for k, v in c.items()})
t = glob.glob('gpurun_out/prof_r05_clock/pmc/**/*kernel_trace.csv', recursive=True)[0]
for r in list(csv.DictReader(open(t)))[-3:]:
    print(r['Kernel_Name'][:30], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 'ms', r.get('Dispatch_Id'))
PY
find $OUT -name "*.csv" -size +5M -delete
