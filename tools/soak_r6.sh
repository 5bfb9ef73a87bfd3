#!/bin/bash
# Round 6 soak runs (evidence of run-to-run determinism; outputs -> gpurun_out/soak6/, copied to profiles/r6/soak_*.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/soak6
mkdir -p $OUT
cd $R
{ for a in "1024 3000" "2048 2000" "4608 600" "8192 400"; do set -- $a; timeout 1500 python3 tools/chain_stress.py $1 $2 2>&1 | tail -1; done; } > $OUT/soak_chain.txt
{ for a in "1024 1500" "4608 300" "8192 200"; do set -- $a; timeout 1500 python3 tools/doubling_repeat.py $1 $2 2>&1 | tail -1; done; } > $OUT/soak_factor_solve.txt
{ timeout 1500 python3 tools/lockstep_split_soak.py 64 2048 150 2>&1 | tail -1; timeout 900 python3 tools/lockstep_split_soak.py 32 2048 200 2>&1 | tail -1; timeout 900 python3 tools/lockstep_split_soak.py 16 2048 200 2>&1 | tail -1; } > $OUT/soak_lockstep_split.txt
{ echo "== order 256"; timeout 1800 python3 tools/nt1024_repeat.py 256 20000 6 2>/dev/null | tail -1; echo "== order 400 (padded 512)"; timeout 900 python3 tools/nt1024_repeat.py 400 2000 6 2>/dev/null | tail -1; echo "== order 1000 (padded 1024)"; timeout 1500 python3 tools/nt1024_repeat.py 1000 1000 6 2>/dev/null | tail -1; } > $OUT/soak_jacobi.txt
cat $OUT/soak_chain.txt $OUT/soak_factor_solve.txt $OUT/soak_lockstep_split.txt $OUT/soak_jacobi.txt
