#!/bin/bash
# kernel-trace statistics of the headline bench for one or more library builds: per-kernel averages of the panel chain
# usage (GPU box, repo root): bash tools/prof_panel.sh name=path.so ...   ("default" = the in-tree build)
R=$(cd "$(dirname "$0")/.." && pwd); OUT=$R/gpurun_out/prof_panel; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for a in "$@"; do
  name=${a%%=*}; path=${a#*=}
  [ "$path" = default ] && unset CIPKKT_LIB || export CIPKKT_LIB=$R/$path
  rm -rf /tmp/pp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$name -o t -- python3 $R/bench.py --no-cpu-baseline --no-c5 --no-converge --no-secondary --no-plugin-boundary --steps 10 --warmup 2 > $OUT/$name.json 2> $OUT/$name.err
  f=$(find /tmp/pp_$name -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/${name}_kernel_stats.csv
  echo "== $name"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-60s calls %6s  avg %9.2f us  total %9.3f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
done
