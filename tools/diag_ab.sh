#!/bin/bash
# A/B of the diagonal kernel's step A on the GPU box: round 2's pivot-by-pivot form (-DDIAG_STEP_A_REF) against the
# rank-4 blocked form -- time per launch and bit-identity of L, d and the micro inverses on three matrix classes.
# usage (on the GPU box, from the repo root): bash tools/diag_ab.sh [outdir]
# the three binaries are built in the build container (they travel with the snapshot):
#   cd tools && for v in new:'' ref:-DDIAG_STEP_A_REF tim:-DDIAG_TIMING; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -w -I../include ${v#*:} -o diag_bench_${v%%:*} diag_bench.hip; done
set -u
R=$(cd "$(dirname "$0")/.." && pwd); OUT=${1:-$R/gpurun_out/diag_ab}; mkdir -p $OUT
for m in 0 1 2; do
  for v in ref new; do
    DIAG_BENCH_MATRIX=$m DIAG_BENCH_DUMP=$OUT/dump_${v}_$m.bin $R/tools/diag_bench_$v > $OUT/${v}_$m.log 2>&1
    [ "$m" = 0 ] && unset DIAG_BENCH_MATRIX
  done
  if cmp -s $OUT/dump_ref_$m.bin $OUT/dump_new_$m.bin; then echo "matrix $m: BIT-IDENTICAL"; else echo "matrix $m: DIFFERENT"; fi
  tail -2 $OUT/ref_$m.log | sed "s/^/  ref: /"; tail -2 $OUT/new_$m.log | sed "s/^/  new: /"
done
[ -x $R/tools/diag_bench_tim ] && $R/tools/diag_bench_tim | tail -9
rm -f $OUT/dump_*.bin
