#!/bin/bash
# Collects the judged profile artefacts on the GPU box (run through gpurun from the repo root):
#   the default bench line, kernel-trace stats of the same command (headline workload only: --no-c5, no CPU leg, no
#   convergence run, so that the per-kernel averages are those of the timed steps), three separate PMC passes (no
#   tracing with --pmc), and kernel stats of configs 3 and 4.
# Outputs land in gpurun_out/final/; copy the summaries into profiles/<round>/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o final -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-converge --no-c5 > $OUT/bench_profiled.json 2>/dev/null
cp $OUT/stats/final_kernel_stats.csv $OUT/final_kernel_stats.csv
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-converge --no-c5 > /dev/null 2>&1
  python3 $R/tools/pmc_extract.py $OUT/pmc_$name/pmc_counter_collection.csv $OUT/final_pmc_$name.csv
done
# config 4 (SDP, matrix order 256) and config 3 (SOCP, 512 x Q(8)): per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 $R/tools/c4_time.py 256 > $OUT/c4_time.log 2>/dev/null
cp $OUT/c4/c4_kernel_stats.csv $OUT/c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python3 $R/tools/bench_configs.py c3 > $OUT/c3_time.log 2>/dev/null
cp $OUT/c3/c3_kernel_stats.csv $OUT/c3_kernel_stats.csv
# the opt-in look-ahead schedule: timing + kernel trace summary
python3 $R/tools/la_time.py 8192 01 > $OUT/lookahead_time.log 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d $OUT/la -o la -- python3 $R/tools/la_time.py 8192 1 > /dev/null 2>&1
python3 $R/tools/la_trace.py $OUT/la > $OUT/lookahead_trace_summary.txt 2>&1
CIP_LA_DBG=64 python3 $R/tools/la_time.py 8192 1 > $OUT/lookahead_chain_only.log 2>/dev/null
# the opt-in two-stream look-ahead: timing (bit-identity checked in the same run) + overlap timeline
python3 $R/tools/la2_time.py 8192 > $OUT/lookahead2_time.log 2>/dev/null
CIP_LOOKAHEAD=3 rocprofv3 --kernel-trace --output-format csv -d $OUT/la2 -o la2 -- python3 $R/tools/la2_time.py 8192 > /dev/null 2>&1
python3 $R/tools/la2_trace.py $OUT/la2 > $OUT/lookahead2_trace_summary.txt 2>&1
# the panel chain's three forms: three launches per panel / diag + in-block update / one launch per panel with the TRSM
# pipelined behind the diagonal kernel (default); the diagonal kernel with 4 / 8 / 12 waves
for f in 0 1 3; do echo "CIP_FUSE_DIAG=$f"; CIP_FUSE_DIAG=$f python3 $R/tools/la_time.py 8192 0 2>/dev/null | tail -1; done > $OUT/fused_chain_time.log
for w in 4 8 12; do echo "CIP_DIAG_WAVES=$w (standalone diagonal kernel, chain form 0)"; CIP_FUSE_DIAG=0 CIP_DIAG_WAVES=$w python3 $R/tools/la_time.py 8192 0 2>/dev/null | tail -1; done >> $OUT/fused_chain_time.log
# config 5 on one GPU (the multi-GPU workload of bench.py): lock-step (default) and the thread pool
python3 $R/bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> /dev/null
python3 $R/bench.py --workload c5 --batch-mode threads --steps 3 --warmup 1 > $OUT/bench_c5_threads.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5ls -o c5ls -- python3 $R/tools/lockstep_time.py 64 2048 1 lockstep > $OUT/c5_lockstep_time.log 2>&1
cp $OUT/c5ls/c5ls_kernel_stats.csv $OUT/c5_lockstep_kernel_stats.csv
for c in 8 16 32 64; do CIP_LOCKSTEP_TIMING=1 python3 $R/tools/lockstep_time.py $c 2048 2 both 2>&1 | tail -5; done > $OUT/c5_shard_sizes.log
ls -la $OUT
