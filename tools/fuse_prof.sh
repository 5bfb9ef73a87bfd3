cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in ${FUSE_LIST:-0 1}; do
CIP_FUSE_DIAG=$f rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fuse$f -o f -- python3 $R/tools/la_time.py ${1:-2048} 0 > /dev/null 2>&1
echo "fuse $f"
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/fuse$f/f_kernel_stats.csv')))
for r in rows:
    if r['Name'].startswith(('k_ldlt','k_trsm','k_gemm_nt_64(','void k_ldlt')):
        print('%-50s %6d avg %8.1f us total %8.2f ms'%(r['Name'][:50], int(r['Calls']), float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
