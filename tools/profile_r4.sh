#!/bin/bash
# Collects the judged profile artefacts of round 4 on the GPU box (run through gpurun from the repo root); outputs land in
# gpurun_out/final4/, the summaries are copied into profiles/r4/ afterwards.  Same structure as tools/profile_r3.sh plus: the
# literal 3x3 route at the headline size (N = 16384), the large-SOC timing, the side-stream A/B, the tail timeline of a
# factorisation (both streams), the 8-problem shard's kernel statistics and the machine-readable rooflines of configs 3 / 4 / 5.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/final_bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o final -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-converge --no-c5 > $OUT/final_bench_profiled.json 2>/dev/null
cp $OUT/stats/final_kernel_stats.csv $OUT/final_kernel_stats.csv
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-converge --no-c5 > /dev/null 2>&1
  python3 $R/tools/pmc_extract.py $OUT/pmc_$name/pmc_counter_collection.csv $OUT/final_pmc_$name.csv
done
# the literal 3x3 route at the headline size: N = n + m = 16384 (src/kktsolvers.jl:254-257)
python3 $R/bench.py --route full3x3 --steps 5 --warmup 2 --no-cpu-baseline --no-c5 > $OUT/bench_full3x3.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f33 -o f33 -- python3 $R/bench.py --route full3x3 --steps 3 --warmup 1 --no-cpu-baseline --no-converge --no-c5 > /dev/null 2>&1
cp $OUT/f33/f33_kernel_stats.csv $OUT/full3x3_kernel_stats.csv
# per-launch durations along one factorisation; the end of a factorisation on both streams (side-stream solve preparation on / off)
bash $R/tools/panel_trace.sh r4=default > $OUT/panel_trace.txt 2>&1
bash $R/tools/tail_trace.sh CIP_SIDE_PREP=1 > /dev/null 2>&1; cp $R/gpurun_out/tail_trace/CIP_SIDE_PREP_1.txt $OUT/tail_trace_side_on.txt
bash $R/tools/tail_trace.sh CIP_SIDE_PREP=0 > /dev/null 2>&1; cp $R/gpurun_out/tail_trace/CIP_SIDE_PREP_0.txt $OUT/tail_trace_side_off.txt
bash $R/tools/step_trace.sh r4=default > $OUT/step_trace.txt 2>&1
# same-session A/B: side-stream preparation off; the mirror pass on top of the L' stores (what the pass used to cost); and, when
# the variant library travels with the snapshot (tools/build_variant.sh notstore diag.hip -DCIP_NO_TSTORE), round 3's behaviour
AB="new=default noside=env:CIP_SIDE_PREP=0 plusmirror=env:CIP_LDLT_MIRROR=1"
python3 $R/tools/ab_factor.py $AB --rounds 3 > $OUT/ab_side_mirror.txt 2>&1
# the diagonal kernel alone; solve4x4
bash $R/tools/diag_ab.sh $OUT/diag_ab > $OUT/diag_ab.txt 2>&1
python3 $R/tools/solve_time.py > $OUT/solve_time.txt 2>&1
# large second-order cones
python3 $R/tools/soc_large.py > $OUT/soc_large.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/soc -o soc -- python3 $R/tools/soc_large.py > /dev/null 2>&1
cp $OUT/soc/soc_kernel_stats.csv $OUT/soc_large_kernel_stats.csv
# configs 4 and 3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 $R/tools/c4_time.py 256 > $OUT/c4_time.txt 2>/dev/null
cp $OUT/c4/c4_kernel_stats.csv $OUT/c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python3 $R/tools/bench_configs.py c3 > $OUT/c3_time.txt 2>/dev/null
cp $OUT/c3/c3_kernel_stats.csv $OUT/c3_kernel_stats.csv
# config 5 on one GPU: lock-step (default), the RCCL path with one rank, shard sizes, the 8-problem shard's kernels
python3 $R/bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> /dev/null
CIP_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29741 $R/bench.py --gpus 1 --workload c5 --steps 3 --warmup 1 > $OUT/bench_c5_rccl_one_rank.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5ls -o c5ls -- python3 $R/tools/lockstep_time.py 64 2048 1 lockstep > $OUT/c5_lockstep_time.txt 2>&1
cp $OUT/c5ls/c5ls_kernel_stats.csv $OUT/c5_lockstep_kernel_stats.csv
for c in 8 16 32 64; do CIP_LOCKSTEP_TIMING=1 python3 $R/tools/lockstep_time.py $c 2048 2 both 2>&1 | tail -5; done > $OUT/c5_shard_sizes.txt
bash $R/tools/prof_lockstep.sh 8 2048 $OUT/c5b8 > $OUT/c5_b8_profile.txt 2>&1
cp $OUT/c5b8/b8_kernel_stats.csv $OUT/c5_b8_kernel_stats.csv
# per-iteration kernel timelines (launches, kernel time, idle time, gaps): config 4 and the 8-problem lock-step shard
python3 $R/tools/loop_run.py c4 > $OUT/c4_loop_time.txt 2>/dev/null
rm -rf /tmp/lr_c4; rocprofv3 --kernel-trace --output-format csv -d /tmp/lr_c4 -o t -- python3 $R/tools/loop_run.py c4 > /dev/null 2>&1
python3 $R/tools/iter_timeline.py $(find /tmp/lr_c4 -name "*kernel_trace.csv" | head -1) k_lg_jacobi 30 > $OUT/c4_iter_timeline.txt 2>&1
bash $R/tools/lockstep_trace.sh 8 2048 $OUT/lt8 > /dev/null 2>&1; cp $OUT/lt8/b8_timeline.txt $OUT/c5_b8_iter_timeline.txt
python3 $R/tools/config_rooflines.py $OUT > $OUT/rooflines.json
rm -rf $OUT/stats $OUT/pmc_* $OUT/c4 $OUT/c3 $OUT/c5ls $OUT/f33 $OUT/soc $OUT/c5b8 $OUT/diag_ab $OUT/lt8
ls -la $OUT
