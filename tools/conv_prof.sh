cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/conv -o c -- python3 $R/bench.py --no-cpu-baseline --no-c5 --steps 1 --warmup 0 > /dev/null 2>&1
python3 - <<'PY'
import csv,os
R=os.environ['GRAFT_REPO_ROOT']
rows=list(csv.DictReader(open(R+'/gpurun_out/conv/c_kernel_stats.csv')))
tot=0
for r in rows:
    n=r['Name']
    if n.startswith('void at::') or n.startswith('Cijk') or 'elementwise' in n: continue
    tot+=float(r['TotalDurationNs'])
print('own kernels total ms', tot/1e6)
for r in rows[:40]:
    n=r['Name']
    if n.startswith('void at::') or n.startswith('Cijk') or 'elementwise' in n: continue
    print('%-58s %6d %9.2f ms %8.1f us'%(n[:58], int(r['Calls']), float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
