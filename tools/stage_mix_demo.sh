#!/bin/bash
# The stage-count defect of the fused panel chain (rounds 5 - 6a) made visible: with the helper waves' write-back of micro-panel 6 held
# back by 2000 ticks of s_memtime (what a slow write-through store does once in ~100 000 factorisations of order 4096) the OLD counting hands the TRSM
# strips stage 6 before its data -- the factor changes; the counting of round 6b does not care.  Needs a GPU.
#   tools/stage_mix_demo.sh [seconds per leg]
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
T=${1:-5}
bash tools/build_variant.sh newslow diag.hip "-DDIAG_DEBUG_SLOW_HELPERS=2000" >/dev/null
bash tools/build_variant.sh oldslow diag.hip "-DDIAG_DEBUG_SLOW_HELPERS=2000 -DDIAG_OLD_STAGE_COUNT" >/dev/null
bash tools/build_variant.sh oldcount diag.hip "-DDIAG_OLD_STAGE_COUNT" >/dev/null
V=$R/conicip.jl_amd/build/variants
for leg in "library: " "newslow:$V/libcipkkt_newslow.so" "oldcount:$V/libcipkkt_oldcount.so" "oldslow:$V/libcipkkt_oldslow.so"; do
  name=${leg%%:*}; lib=${leg#*:}
  echo "== $name"
  if [ -n "$lib" ] && [ "$lib" != " " ]; then export CIPKKT_LIB=$lib; else unset CIPKKT_LIB; fi
  timeout 300 python3 tools/kkt_soak.py full3x3 2048 $T 2>&1 | grep -v amdgpu.ids | grep "reference solutions\|factorisations\|ODD" | head -6
done
