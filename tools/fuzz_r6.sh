#!/bin/bash
# Round 6 fuzz runs on the final library (outputs -> gpurun_out/fuzz6/, copied to profiles/r6/fuzz_*.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fuzz6
mkdir -p $OUT
cd $R
timeout 2400 python3 tools/fuzz_status.py 1200 > $OUT/fuzz_status.txt 2>&1
tail -3 $OUT/fuzz_status.txt
timeout 1500 python3 tools/fuzz_sdp.py 6 150,200,256,300,450,640 > $OUT/fuzz_sdp_large.txt 2>&1
tail -8 $OUT/fuzz_sdp_large.txt
