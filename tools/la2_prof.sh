cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CIP_LOOKAHEAD=3 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/la2 -o la2 -- python3 $R/tools/la2_time.py 8192 > $R/gpurun_out/la2_time.log 2>&1
python3 $R/tools/la2_trace.py $R/gpurun_out/la2
