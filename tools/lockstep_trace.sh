#!/bin/bash
# kernel timeline of the interior-point iterations of ONE lock-step pass over B problems of order n (rocprofv3 kernel trace):
# launches, kernel time and idle time per iteration, by kernel and by (kernel -> next kernel) gap.
# usage: bash tools/lockstep_trace.sh B [n] [outdir]
R=$(cd "$(dirname "$0")/.." && pwd); B=${1:-8}; N=${2:-2048}; OUT=${3:-$R/gpurun_out/lockstep_trace}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/lt_$B
rocprofv3 --kernel-trace --output-format csv -d /tmp/lt_$B -o t -- python3 $R/tools/lockstep_time.py $B $N 2 lockstep > $OUT/b$B.out 2> $OUT/b$B.err
f=$(find /tmp/lt_$B -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/b${B}_timeline.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
idx = [i for i, r in enumerate(rows) if nm(r).startswith("k_nt_scaling")]
# the last pass: the trailing run of k_nt_scaling launches less than 5 ms apart
run = [idx[-1]]
for i in reversed(idx[:-1]):
    if int(rows[run[0]]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) < 5e6: run.insert(0, i)
    else: break
a, b = run[0], run[-1]
its = len(run) - 1
seg = rows[a:b]
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print("%d iterations: %.3f ms each, %.1f launches each" % (its, span / its / 1e6, len(seg) / its))
tot = collections.defaultdict(lambda: [0, 0.0]); gg = collections.defaultdict(lambda: [0, 0.0]); busy = 0.0; idle = 0.0
for k, r in enumerate(seg):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[nm(r)][0] += 1; tot[nm(r)][1] += d; busy += d
    g = int(rows[a + k + 1]["Start_Timestamp"]) - int(r["End_Timestamp"])
    if g > 0: idle += g
    key = (nm(r)[:28], nm(rows[a + k + 1])[:28]); gg[key][0] += 1; gg[key][1] += g
print("per iteration: kernel time %.3f ms, idle between launches %.3f ms" % (busy / its / 1e6, idle / its / 1e6))
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %-44s %6.1f x %8.1f us = %8.3f ms" % (n[:44], c / its, t / c / 1e3, t / its / 1e6))
print("idle by (kernel -> next kernel), per iteration:")
for (x, y), (c, t) in sorted(gg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %-28s -> %-28s %5.1f x %6.2f us = %7.3f ms" % (x, y, c / its, t / c / 1e3, t / its / 1e6))
PY
