#!/bin/bash
# kernel timeline of the END of one factorisation of the headline step (from the last trailing update on), both streams, from a
# rocprofv3 kernel trace.   usage: bash tools/tail_trace.sh [VAR=VALUE ...]   (environment of the traced run)
R=$(cd "$(dirname "$0")/.." && pwd); OUT=$R/gpurun_out/tail_trace; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
tag=$(echo "$*" | tr ' =' '__'); [ -z "$tag" ] && tag=default
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/tt_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tt_$tag -o t -- python3 $R/bench.py --no-cpu-baseline --no-c5 --no-converge --steps 4 --warmup 2 > $OUT/$tag.out 2> $OUT/$tag.err
f=$(find /tmp/tt_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/$tag.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
tr = [i for i, r in enumerate(rows) if nm(r).startswith("k_ldlt_trailing")]
# the 7th trailing update of the last-but-one factorisation of the run
groups = [tr[i:i + 7] for i in range(0, len(tr) - 6, 7)]
g = groups[-2]
lo = g[-1]
nt = [i for i in range(lo, len(rows)) if nm(rows[i]).startswith("k_nt_scaling") or nm(rows[i]).startswith("k_s4_pre")]
hi = nt[0] + 40 if nt else min(len(rows), lo + 80)
t0 = int(rows[lo]["Start_Timestamp"])
for i in range(lo, hi):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %7.1f  q%-3s grid %8s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Grid_Size_X"], nm(r)[:40]))
PY
