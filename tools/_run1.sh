mkdir -p gpurun_out/loops
cd /tmp; export TMPDIR=/tmp
for c in c2 c3; do
  rm -rf /tmp/lr_$c
  rocprofv3 --kernel-trace --output-format csv -d /tmp/lr_$c -o t -- python3 $GRAFT_REPO_ROOT/tools/loop_run.py $c > $GRAFT_REPO_ROOT/gpurun_out/loops/$c.out 2>&1
  f=$(find /tmp/lr_$c -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/iter_timeline.py $f k_nt_scaling 30 > $GRAFT_REPO_ROOT/gpurun_out/loops/${c}_timeline.txt 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/loop_run.py c2 > $GRAFT_REPO_ROOT/gpurun_out/loops/c2_plain.out 2>&1
python3 $GRAFT_REPO_ROOT/tools/loop_run.py c3 > $GRAFT_REPO_ROOT/gpurun_out/loops/c3_plain.out 2>&1
