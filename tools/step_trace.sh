#!/bin/bash
# kernel timeline of ONE headline step (NT scaling + assembly + LDL' + 2 solve4x4, n = 8192) from a rocprofv3 kernel trace:
# per-kernel totals, the idle time between launches, the largest gaps.   usage: bash tools/step_trace.sh [name=path.so]
R=$(cd "$(dirname "$0")/.." && pwd); OUT=$R/gpurun_out/step_trace; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
a=${1:-new=default}; name=${a%%=*}; path=${a#*=}
[ "$path" = default ] && unset CIPKKT_LIB || export CIPKKT_LIB=$R/$path
rm -rf /tmp/st_$name
rocprofv3 --kernel-trace --output-format csv -d /tmp/st_$name -o t -- python3 $R/bench.py --no-cpu-baseline --no-c5 --steps 4 --warmup 2 > $OUT/$name.out 2> $OUT/$name.err
f=$(find /tmp/st_$name -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/$name.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
# steps of the timed loop: each begins with k_nt_scaling; the timed loop's steps are the last run of equally spaced ones
idx = [i for i, r in enumerate(rows) if nm(r).startswith("k_nt_scaling")]
# pick a step in the middle of the timed region: the 6 steps (2 warm-up + 4) come before check_factor / the split run
runs = []
for a, b in zip(idx, idx[1:]):
    runs.append((a, b, int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])))
cand = [x for x in runs if 4.5e6 < x[2] < 7e6 and not any(nm(r).startswith("k_maxstep") for r in rows[x[0]:x[1]])]   # bench steps, not IPM iterations
a, b, span = cand[len(cand) // 2]
seg = rows[a:b]
print("step span %.3f ms, %d launches" % (span / 1e6, len(seg)))
tot = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0; gaps = []
for k, r in enumerate(seg):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[nm(r)][0] += 1; tot[nm(r)][1] += d
    nxt = int(seg[k + 1]["Start_Timestamp"]) if k + 1 < len(seg) else int(rows[b]["Start_Timestamp"])
    g = nxt - int(r["End_Timestamp"])
    gaps.append((g, nm(r), nm(seg[k + 1]) if k + 1 < len(seg) else "next step"))
    busy += d
print("sum of kernel durations %.3f ms, idle between launches %.3f ms" % (busy / 1e6, sum(g for g, _, _ in gaps if g > 0) / 1e6))
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %-44s %4d x %8.1f us = %8.3f ms" % (n[:44], c, t / c / 1e3, t / 1e6))
gg = collections.defaultdict(lambda: [0, 0.0])
for g, x, y in gaps:
    gg[(x[:28], y[:28])][0] += 1; gg[(x[:28], y[:28])][1] += g
print("idle by (kernel -> next kernel):")
for (x, y), (c, t) in sorted(gg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-28s -> %-28s %4d x %6.2f us = %7.3f ms" % (x, y, c, t / c / 1e3, t / 1e6))
PY
