cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ls8 -o ls -- python3 $R/tools/lockstep_time.py ${1:-8} 2048 1 lockstep > $R/gpurun_out/ls8_time.log 2>&1
tail -2 $R/gpurun_out/ls8_time.log
python3 - <<'PY'
import csv,os
R=os.environ['GRAFT_REPO_ROOT']
rows=list(csv.DictReader(open(R+'/gpurun_out/ls8/ls_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows if not r['Name'].startswith('void at::') and not r['Name'].startswith('Cijk'))
print('own kernels total ms', tot/1e6)
for r in rows[:30]:
    if r['Name'].startswith('void at::') or r['Name'].startswith('Cijk'): continue
    print('%-60s %7d %9.2f ms %8.1f us'%(r['Name'][:60], int(r['Calls']), float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
