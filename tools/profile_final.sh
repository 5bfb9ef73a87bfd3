#!/bin/bash
# Collects the judged profile artefacts on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench command, then three separate PMC passes (no tracing with --pmc).
# Outputs land in gpurun_out/final/; copy the summaries into profiles/<round>/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o final -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-converge > /dev/null 2>&1
cp $OUT/stats/final_kernel_stats.csv $OUT/final_kernel_stats.csv
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-converge > /dev/null 2>&1
  python3 $R/tools/pmc_extract.py $OUT/pmc_$name/pmc_counter_collection.csv $OUT/final_pmc_$name.csv
done
ls -la $OUT
