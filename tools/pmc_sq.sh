#!/bin/bash
# SQ stall / LDS counters of the LDL' kernels at the headline size (two separate --pmc passes, counters only), summarised per kernel:
# wave cycles split into parked (s_waitcnt / barrier), issue-stalled and issuing; LDS bank-conflict cycles against LDS-active cycles.
# usage: bash tools/pmc_sq.sh [outdir] [workload script + args, default: bench.py at the headline size]   (every kernel with >= 0.5 % of the wave cycles is listed)
R=$(cd "$(dirname "$0")/.." && pwd); OUT=${1:-$R/gpurun_out/pmc_sq}; shift; WL=${*:-$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-converge --no-c5 --no-live-pmc}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for pass in "stall:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf /tmp/pmcsq_$name
  rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmcsq_$name -o pmc -- python3 $WL > /dev/null 2>&1
  cp $(find /tmp/pmcsq_$name -name "*counter_collection.csv" | head -1) $OUT/$name.csv
done
python3 - $OUT <<'PY' | tee $OUT/pmc_sq_summary.txt
import csv, sys, collections
out = sys.argv[1]
for name in ("stall", "lds"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open("%s/%s.csv" % (out, name))):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen: seen.add((r["Dispatch_Id"], k)); cnt[k] += 1
    print("== pass %s (sums over all dispatches of the run; SQ_* cycle counters count quad-cycles)" % name)
    tot = sum(c.get("SQ_WAVE_CYCLES", 0.0) for c in acc.values()) or 1.0
    for k in sorted(acc, key=lambda k: -(acc[k].get("SQ_WAVE_CYCLES", 0.0) + acc[k].get("SQ_LDS_IDX_ACTIVE", 0.0))):
        c = acc[k]
        if name == "stall" and c.get("SQ_WAVE_CYCLES", 0.0) < 0.005 * tot: continue
        if name == "lds" and not c.get("SQ_LDS_IDX_ACTIVE"): continue
        line = "%-28s %5d dispatches " % (k[:28], cnt[k])
        if name == "stall" and c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            line += "| of wave cycles: parked %.1f %%, issue-stalled %.1f %% (LDS-issue %.1f %%), issuing %.1f %%" % (100 * c["SQ_WAIT_ANY"] / w, 100 * c["SQ_WAIT_INST_ANY"] / w, 100 * c["SQ_WAIT_INST_LDS"] / w, 100 * c["SQ_ACTIVE_INST_ANY"] / w)
        if name == "lds" and c.get("SQ_LDS_IDX_ACTIVE"):
            line += "| LDS bank-conflict cycles / LDS-active cycles = %.3f, LDS instructions %.3g" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], c.get("SQ_INSTS_LDS", 0))
        print(line)
PY
rm -f $OUT/stall.csv $OUT/lds.csv
