#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_att20 -- python3 $GRAFT_REPO_ROOT/tools/experiments/run_attempt_n.py 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_att20.log 2>&1
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_att20/*/*kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
rows=rows[-40:]
prev=None
for r in rows:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-42s dur %7.1f us  gap %7.1f us'%(r['Kernel_Name'].split('(')[0][-42:], (en-st)/1e3, ((st-prev)/1e3) if prev else 0))
    prev=en
PY
