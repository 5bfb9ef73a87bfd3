#!/bin/bash
# Collects the judged profile artefacts of round 3 on the GPU box (run through gpurun from the repo root):
#   the default bench line; kernel-trace stats of the same command (headline workload only: no CPU leg, no convergence run,
#   no config-5 figure, so that the per-kernel averages are those of the timed steps); three separate PMC passes (no tracing
#   with --pmc); the trailing update's XCD-patch A/B; per-launch durations along one factorisation; configs 3, 4, 5.
# Outputs land in gpurun_out/final/; copy the summaries into profiles/r3/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 > $OUT/final_bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o final -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-converge --no-c5 > $OUT/final_bench_profiled.json 2>/dev/null
cp $OUT/stats/final_kernel_stats.csv $OUT/final_kernel_stats.csv
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-converge --no-c5 > /dev/null 2>&1
  python3 $R/tools/pmc_extract.py $OUT/pmc_$name/pmc_counter_collection.csv $OUT/final_pmc_$name.csv
done
# XCD-aware patch order of the trailing update (CIP_TRAIL_PATCH = 4 / 8) against the plain tile order, same session, 3 rounds:
# time, and the L2-miss traffic of each order
python3 $R/tools/ab_factor.py plain=default patch4=env:CIP_TRAIL_PATCH=4 patch8=env:CIP_TRAIL_PATCH=8 --rounds 3 > $OUT/trail_patch_ab.txt 2>&1
for v in 4 8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    CIP_TRAIL_PATCH=$v rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_patch${v}_$c -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-converge --no-c5 > /dev/null 2>&1
    python3 $R/tools/pmc_extract.py $OUT/pmc_patch${v}_$c/pmc_counter_collection.csv $OUT/patch${v}_pmc_$c.csv
  done
done
python3 $R/tools/pmc_traffic.py $OUT/final_pmc_fetch.csv $OUT/final_pmc_write.csv plain >> $OUT/trail_patch_ab.txt 2>&1
for v in 4 8; do python3 $R/tools/pmc_traffic.py $OUT/patch${v}_pmc_FETCH_SIZE.csv $OUT/patch${v}_pmc_WRITE_SIZE.csv patch$v >> $OUT/trail_patch_ab.txt 2>&1; done
# per-launch durations of the panel chain and the trailing updates along one factorisation
bash $R/tools/panel_trace.sh r3=default > $OUT/panel_trace.txt 2>&1
# phase stamps of one panel launch (library built with -DPANEL_TIMING: tools/build_variant.sh ptim diag.hip -DPANEL_TIMING): a quiet
# launch at the bottom of the matrix and the tile-bound second launch of the factorisation; the whole step's kernel timeline
if [ -f $R/conicip.jl_amd/build/variants/libcipkkt_ptim.so ]; then
  ( echo "n = 2048, last panel launch with an update:"; CIPKKT_LIB=$R/conicip.jl_amd/build/variants/libcipkkt_ptim.so python3 $R/tools/panel_stamps.py 2048 2>&1 | tail -2
    echo "n = 8192, the launch of the panel at column 128 (964 update tiles):"; CIPKKT_LIB=$R/conicip.jl_amd/build/variants/libcipkkt_ptim.so python3 $R/tools/panel_stamps.py 8192 128 2>&1 | tail -2 ) > $OUT/panel_stamps.txt
fi
bash $R/tools/step_trace.sh r3=default > $OUT/step_trace.txt 2>&1
# the diagonal kernel: round 2's step A / helper schedule against round 3's (bit-identity + time per kernel + phase clocks)
bash $R/tools/diag_ab.sh $OUT/diag_ab > $OUT/diag_ab.txt 2>&1
# solve4x4: fused element-wise kernels around the sweeps against the separate launches
( python3 $R/tools/solve_time.py; echo "CIP_S4_FUSED=0:"; CIP_S4_FUSED=0 python3 $R/tools/solve_time.py ) > $OUT/solve_time.txt 2>&1
# config 4 (SDP, matrix order 256) and config 3 (SOCP, 512 x Q(8)): per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 $R/tools/c4_time.py 256 > $OUT/c4_time.txt 2>/dev/null
cp $OUT/c4/c4_kernel_stats.csv $OUT/c4_kernel_stats.csv
( CIP_LG_LANCZOS_STATS=1 python3 $R/tools/c4_run.py 256; echo "CIP_LG_LANCZOS=0:"; CIP_LG_LANCZOS=0 python3 $R/tools/c4_run.py 256 ) > $OUT/c4_lanczos_ab.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python3 $R/tools/bench_configs.py c3 > $OUT/c3_time.txt 2>/dev/null
cp $OUT/c3/c3_kernel_stats.csv $OUT/c3_kernel_stats.csv
# config 5 on one GPU (the multi-GPU workload of bench.py): lock-step (default), the RCCL path with one rank, shard sizes
python3 $R/bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> /dev/null
CIP_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29741 $R/bench.py --gpus 1 --workload c5 --steps 3 --warmup 1 > $OUT/bench_c5_rccl_one_rank.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5ls -o c5ls -- python3 $R/tools/lockstep_time.py 64 2048 1 lockstep > $OUT/c5_lockstep_time.txt 2>&1
cp $OUT/c5ls/c5ls_kernel_stats.csv $OUT/c5_lockstep_kernel_stats.csv
for c in 8 16 32 64; do CIP_LOCKSTEP_TIMING=1 python3 $R/tools/lockstep_time.py $c 2048 2 both 2>&1 | tail -5; done > $OUT/c5_shard_sizes.txt
rm -rf $OUT/stats $OUT/pmc_* $OUT/c4 $OUT/c3 $OUT/c5ls
ls -la $OUT
