cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ls -o ls -- python3 $R/tools/lockstep_time.py 64 2048 1 lockstep > $R/gpurun_out/ls_time.log 2>&1
python3 - <<'PY'
import csv,os
R=os.environ['GRAFT_REPO_ROOT']
rows=list(csv.DictReader(open(R+'/gpurun_out/ls/ls_kernel_stats.csv')))
for r in rows[:28]:
    print('%-60s %7d %9.2f ms %8.1f us %5.1f%%'%(r['Name'][:60], int(r['Calls']), float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
