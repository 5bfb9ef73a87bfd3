#!/bin/bash
# SQ stall / LDS counters of the LDL' kernels at the headline size (two separate --pmc passes, counters only), summarised per kernel:
# wave cycles split into parked (s_waitcnt / barrier), issue-stalled and issuing; LDS bank-conflict cycles against LDS-active cycles.
# usage: bash tools/pmc_sq.sh [outdir]
R=$(cd "$(dirname "$0")/.." && pwd); OUT=${1:-$R/gpurun_out/pmc_sq}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for pass in "stall:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf /tmp/pmcsq_$name
  rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmcsq_$name -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-converge --no-c5 --no-live-pmc > /dev/null 2>&1
  cp $(find /tmp/pmcsq_$name -name "*counter_collection.csv" | head -1) $OUT/$name.csv
done
python3 - $OUT <<'PY' | tee $OUT/pmc_sq_summary.txt
import csv, sys, collections
out = sys.argv[1]
KEEP = ("k_ldlt_trailing_64", "k_ldlt_panel", "k_gemv_t", "k_gemm_nt_64_batched", "k_diag_inverse_batched")
for name in ("stall", "lds"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open("%s/%s.csv" % (out, name))):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not k.startswith(KEEP): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen: seen.add((r["Dispatch_Id"], k)); cnt[k] += 1
    print("== pass %s (sums over all dispatches of the run; SQ_* cycle counters count quad-cycles)" % name)
    for k in sorted(acc):
        c = acc[k]
        line = "%-28s %5d dispatches " % (k[:28], cnt[k])
        if name == "stall" and c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            line += "| of wave cycles: parked %.1f %%, issue-stalled %.1f %% (LDS-issue %.1f %%), issuing %.1f %%" % (100 * c["SQ_WAIT_ANY"] / w, 100 * c["SQ_WAIT_INST_ANY"] / w, 100 * c["SQ_WAIT_INST_LDS"] / w, 100 * c["SQ_ACTIVE_INST_ANY"] / w)
        if name == "lds" and c.get("SQ_LDS_IDX_ACTIVE"):
            line += "| LDS bank-conflict cycles / LDS-active cycles = %.3f, LDS instructions %.3g" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], c.get("SQ_INSTS_LDS", 0))
        print(line)
PY
rm -f $OUT/stall.csv $OUT/lds.csv
