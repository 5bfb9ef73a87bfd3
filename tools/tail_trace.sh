#!/bin/bash
# kernel trace of the END of a factorisation and the first solve behind it (n = 8192): every launch from the third-last panel
# launch to the end of the first solve4x4, with its queue -- the side stream's last group beside the forward sweep
R=$(cd "$(dirname "$0")/.." && pwd); cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tt
rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o t -- python3 $R/tools/step_split.py > /dev/null 2>&1
f=$(find /tmp/tt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
pan = [i for i, r in enumerate(rows) if nm(r).startswith("k_ldlt_panel")]
# the last factorisation that is followed by solves: take the last panel launch that has >= 60 kernels behind it
last = [i for i in pan if i + 80 < len(rows)][-1]
while last + 1 < len(rows) and last + 1 in pan: last += 1
t0 = int(rows[last]["End_Timestamp"])
for r in rows[last - 2:last + 75]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("q%-3s %8.1f -> %8.1f  (%6.1f us)  %s  grid %s" % (r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, nm(r), r.get("Grid_Size", "")))
PY
