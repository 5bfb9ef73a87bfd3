#!/bin/bash
# Collects the judged profile artefacts of round 6 on the GPU box (run through gpurun from the repo root); outputs land in
# gpurun_out/final6/, the summaries are copied into profiles/r6/ afterwards.  Structure of tools/profile_r5.sh.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final6
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
bash $R/tools/panel_trace.sh r6=default > $OUT/panel_trace.txt 2>&1
python3 $R/tools/solve_time.py > $OUT/solve_time.txt 2>&1
python3 $R/tools/step_split.py > $OUT/step_split.txt 2>/dev/null
# configs 3 and 4: loop times, kernel statistics
python3 $R/tools/c4_iter.py > $OUT/c4_iter.txt 2>/dev/null
CIP_LG_DEBUG=1 python3 $R/tools/c4_iter.py 2>&1 | grep -E "rep|jacobi" > $OUT/c4_jacobi_sweeps.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 $R/tools/loop_run.py c4 > $OUT/c4_loop_time.txt 2>/dev/null
cp $OUT/c4/c4_kernel_stats.csv $OUT/c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python3 $R/tools/loop_run.py c3 > $OUT/c3_loop_time.txt 2>/dev/null
cp $OUT/c3/c3_kernel_stats.csv $OUT/c3_kernel_stats.csv
# config 5 on one GPU: lock-step (two groups side by side by default; CIP_LOCKSTEP_SPLIT=1: one after the other), shard sizes, the 8-problem shard's kernels,
# two and eight ranks sharing the GPU
python3 $R/bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> /dev/null
CIP_LOCKSTEP_SPLIT=1 python3 $R/bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_unsplit.json 2> /dev/null
for c in 8 16 32 64; do python3 $R/tools/lockstep_time.py $c 2048 3 lockstep 2>&1 | tail -3; done > $OUT/c5_shard_sizes.txt
bash $R/tools/prof_lockstep.sh 8 2048 $OUT/c5b8 > $OUT/c5_b8_profile.txt 2>&1
cp $OUT/c5b8/b8_kernel_stats.csv $OUT/c5_b8_kernel_stats.csv
CIP_BENCH_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29751 $R/bench.py --gpus 2 --workload c5 --steps 2 --warmup 1 > $OUT/bench_c5_two_ranks_one_gpu.json 2> /dev/null
CIP_BENCH_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29753 $R/bench.py --gpus 8 --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_eight_ranks_one_gpu.json 2> /dev/null
# run-to-run repeatability of a large S cone's NT scaling (one launch per phase: the only form since this round) and of the default panel chain
{ echo "== order 256, 6000 repetitions"; python3 $R/tools/nt1024_repeat.py 256 6000 6 2>/dev/null | tail -2;
  echo "== order 640 (padded 1024), 1000 repetitions"; python3 $R/tools/nt1024_repeat.py 640 1000 6 2>/dev/null | tail -2; } > $OUT/jacobi_repeatability.txt
python3 $R/tools/config_rooflines.py $OUT > $OUT/rooflines.json
rm -rf $OUT/stats $OUT/pmc_* $OUT/c4 $OUT/c3 $OUT/c5b8
ls -la $OUT
