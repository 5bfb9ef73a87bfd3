#!/bin/bash
# rocprofv3 kernel statistics of a lock-step pass over B problems of order n (config-5 shard sizes).  usage: bash tools/prof_lockstep.sh B [n] [outdir]
R=$(cd "$(dirname "$0")/.." && pwd); B=${1:-8}; N=${2:-2048}; OUT=${3:-$R/gpurun_out/prof_lockstep}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pl_$B
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_$B -o t -- python3 $R/tools/lockstep_time.py $B $N 3 lockstep > $OUT/b$B.out 2> $OUT/b$B.err
f=$(find /tmp/pl_$B -name "*kernel_stats.csv" | head -1)
cp $f $OUT/b${B}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms over %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:22]:
    print("%-46s %6s x %8.1f us = %7.2f ms (%4.1f %%)" % (r["Name"].split("(")[0].replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
