#!/bin/bash
# Same-session A/B of the triangular sweeps' block-step forms (CIP_SOLVE_FUSED: 0 two launches per step, 1 one launch for solve
# blocks <= 512, 2 one launch always), run through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-ab_solve}
mkdir -p $OUT
cd $R
B="--steps 20 --warmup 5 --no-cpu-baseline --no-converge --no-c5 --no-secondary --no-plugin-boundary --no-live-pmc"
pick() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d.get('breakdown_ms',{}); print(sys.argv[2], 'value', round(d['value'],2), 'ms/step', round(d['ms_per_step'],4), 'factor', b.get('ldlt_factor'), 'solve4x4', b.get('solve4x4'), 'trail TF', d['roofline']['achieved'])" $1 $2; }
for rep in 1 2; do
  for f in 0 2; do
    CIP_SOLVE_FUSED=$f python3 bench.py $B > $OUT/bench_f${f}_$rep.json 2> $OUT/bench_f${f}_$rep.err; pick $OUT/bench_f${f}_$rep.json "fused=$f rep=$rep"
  done
done
for f in 0 2; do echo "== solve_time fused=$f"; CIP_SOLVE_FUSED=$f python3 tools/solve_time.py 2>&1 | tail -1; done
for f in 0 1; do
  for c in 8 64; do echo "== lockstep $c problems fused=$f"; CIP_SOLVE_FUSED=$f python3 tools/lockstep_time.py $c 2048 3 lockstep 2>&1 | tail -3; done
done
echo "== lockstep 8 problems fused=1 solve block 1024"; CIP_SOLVE_FUSED=2 CIP_LOCKSTEP_SOLVE_BLOCK=1024 python3 tools/lockstep_time.py 8 2048 3 lockstep 2>&1 | tail -3
for f in 0 1; do
  for c in c3 c4; do echo "== $c fused=$f"; CIP_SOLVE_FUSED=$f python3 tools/loop_run.py $c 2>&1 | tail -2; done
done
