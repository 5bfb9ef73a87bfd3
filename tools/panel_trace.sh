#!/bin/bash
# per-launch durations of the panel kernels along ONE factorisation (kernel trace), for one or more library builds
# usage: bash tools/panel_trace.sh name=path.so ...
R=$(cd "$(dirname "$0")/.." && pwd); OUT=$R/gpurun_out/panel_trace; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for a in "$@"; do
  name=${a%%=*}; path=${a#*=}
  [ "$path" = default ] && unset CIPKKT_LIB || export CIPKKT_LIB=$R/$path
  rm -rf /tmp/pt_$name
  rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_$name -o t -- python3 $R/tools/ab_one_factor.py > /dev/null 2> $OUT/$name.err
  f=$(find /tmp/pt_$name -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$name" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST factorisation: from the last k_zero_words before the final run of panel launches
pan = [(i, r) for i, r in enumerate(rows) if "k_ldlt_panel" in r["Kernel_Name"] or "k_ldlt_trailing" in r["Kernel_Name"]]
last = pan[-74:]
out = []
for i, r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tag = "T" if "trailing" in r["Kernel_Name"] else "p"
    out.append("%s%.0f" % (tag, d))
t0 = int(last[0][1]["Start_Timestamp"]); t1 = int(last[-1][1]["End_Timestamp"])
print(sys.argv[2], "span %.3f ms:" % ((t1 - t0) / 1e6), " ".join(out))
PY
done
