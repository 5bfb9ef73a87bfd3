#!/bin/bash
# Collects the judged profile artefacts of round 5 on the GPU box (run through gpurun from the repo root); outputs land in
# gpurun_out/final5/, the summaries are copied into profiles/r5/ afterwards.  Structure of tools/profile_r4.sh, plus: the probes
# behind this round's changes of the diagonal kernel (tools/issue_probe, tools/rcp_probe, tools/diag_bench*), the A/B of the
# fused loop kernels and of the warm-started Jacobi, the two-ranks-on-one-GPU bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/final_bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o final -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-converge --no-c5 --no-secondary --no-plugin-boundary > $OUT/final_bench_profiled.json 2>/dev/null
cp $OUT/stats/final_kernel_stats.csv $OUT/final_kernel_stats.csv
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-converge --no-c5 --no-secondary --no-plugin-boundary > /dev/null 2>&1
  python3 $R/tools/pmc_extract.py $OUT/pmc_$name/pmc_counter_collection.csv $OUT/final_pmc_$name.csv
done
# the diagonal kernel: round 4's (built from the r4 source kept as a variant, when it travels) against this round's; the probes
for b in diag_bench_new diag_bench_tim64 issue_probe rcp_probe; do [ -x $R/tools/$b ] && { echo "== $b"; $R/tools/$b; }; done > $OUT/diag_probes.txt 2>&1
[ -f $R/conicip.jl_amd/build/variants/libcipkkt_r4diag.so ] && python3 $R/tools/ab_factor.py r4diag=$R/conicip.jl_amd/build/variants/libcipkkt_r4diag.so r5=default --rounds 3 > $OUT/ab_diag.txt 2>&1
[ -f $R/conicip.jl_amd/build/variants/libcipkkt_r4diag.so ] && bash $R/tools/prof_panel.sh r4diag=conicip.jl_amd/build/variants/libcipkkt_r4diag.so r5=default > $OUT/prof_panel_ab.txt 2>&1
bash $R/tools/panel_trace.sh r5=default > $OUT/panel_trace.txt 2>&1
python3 $R/tools/solve_time.py > $OUT/solve_time.txt 2>&1
# configs 3 and 4: loop times, kernel statistics; warm-started Jacobi on / off
python3 $R/tools/c4_iter.py > $OUT/c4_iter_warm.txt 2>/dev/null
CIP_LG_WARM=0 python3 $R/tools/c4_iter.py > $OUT/c4_iter_cold.txt 2>/dev/null
CIP_LG_DEBUG=1 python3 $R/tools/c4_iter.py 2>&1 | grep -E "rep|jacobi" > $OUT/c4_jacobi_sweeps_warm.txt
CIP_LG_DEBUG=1 CIP_LG_WARM=0 python3 $R/tools/c4_iter.py 2>&1 | grep -E "rep|jacobi" > $OUT/c4_jacobi_sweeps_cold.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 $R/tools/loop_run.py c4 > $OUT/c4_loop_time.txt 2>/dev/null
cp $OUT/c4/c4_kernel_stats.csv $OUT/c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python3 $R/tools/loop_run.py c3 > $OUT/c3_loop_time.txt 2>/dev/null
cp $OUT/c3/c3_kernel_stats.csv $OUT/c3_kernel_stats.csv
# config 5 on one GPU: lock-step, shard sizes (fused loop kernels on / off), the 8-problem shard's kernels, two ranks sharing the GPU
python3 $R/bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> /dev/null
for c in 8 16 32 64; do python3 $R/tools/lockstep_time.py $c 2048 3 lockstep 2>&1 | tail -3; done > $OUT/c5_shard_sizes.txt
for c in 8 64; do CIP_LOOP_FUSED_R=0 python3 $R/tools/lockstep_time.py $c 2048 3 lockstep 2>&1 | tail -3; done > $OUT/c5_shard_sizes_unfused.txt
for c in 8 64; do CIP_DOTS_FUSED=1 python3 $R/tools/lockstep_time.py $c 2048 3 lockstep 2>&1 | tail -3; done > $OUT/c5_shard_sizes_dots_fused.txt
bash $R/tools/prof_lockstep.sh 8 2048 $OUT/c5b8 > $OUT/c5_b8_profile.txt 2>&1
cp $OUT/c5b8/b8_kernel_stats.csv $OUT/c5_b8_kernel_stats.csv
bash $R/tools/lockstep_trace.sh 8 2048 $OUT/lt8 > /dev/null 2>&1; cp $OUT/lt8/b8_timeline.txt $OUT/c5_b8_iter_timeline.txt
CIP_BENCH_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29751 $R/bench.py --gpus 2 --workload c5 --steps 2 --warmup 1 > $OUT/bench_c5_two_ranks_one_gpu.json 2> /dev/null
# second session of round 5: where a step's time goes on the GPU timeline (first solve behind a factorisation against the second), the end
# of a factorisation + first solve as a kernel trace, same-session A/B against the build at the start of the session, and the
# run-to-run repeatability of a large S cone's NT scaling (persistent against stepped Jacobi)
python3 $R/tools/step_split.py > $OUT/step_split.txt 2>/dev/null
bash $R/tools/tail_trace.sh > $OUT/tail_trace.txt 2>/dev/null
[ -f $R/conicip.jl_amd/build/variants/libcipkkt_base.so ] && bash $R/tools/ab_quick.sh 3 session_start=conicip.jl_amd/build/variants/libcipkkt_base.so final=default > $OUT/ab_session2.txt 2>&1
{ echo "== order 640 (padded 1024), persistent Jacobi (CIP_LG_JACOBI_STEPPED=0)"; CIP_LG_JACOBI_STEPPED=0 python3 $R/tools/nt1024_repeat.py 640 1500 6 2>/dev/null | tail -4;
  echo "== order 640, one launch per phase (default)"; python3 $R/tools/nt1024_repeat.py 640 1000 6 2>/dev/null | tail -2;
  echo "== order 256, persistent (default at this order)"; python3 $R/tools/nt1024_repeat.py 256 3000 6 2>/dev/null | tail -2; } > $OUT/jacobi_repeatability.txt
python3 $R/tools/config_rooflines.py $OUT > $OUT/rooflines.json
rm -rf $OUT/stats $OUT/pmc_* $OUT/c4 $OUT/c3 $OUT/c5b8 $OUT/lt8
ls -la $OUT
