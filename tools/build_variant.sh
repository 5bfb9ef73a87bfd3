#!/bin/bash
# A/B builds of the library: recompile ONE source with extra flags, link it with the other objects of the regular build
# into conicip.jl_amd/build/variants/libcipkkt_<name>.so (select with CIPKKT_LIB=<path>).
# usage: tools/build_variant.sh <name> <source.hip> "<extra flags>"
set -e
R=$(cd "$(dirname "$0")/.." && pwd); B=$R/conicip.jl_amd/build; V=$B/variants; mkdir -p $V
name=$1; src=$2; flags=$3
obj=$V/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -w $flags -c $R/conicip.jl_amd/csrc/$src -o $obj
objs=""
for o in $B/*.o; do [ "$(basename $o)" = "${src%.hip}.o" ] && objs="$objs $obj" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libcipkkt_$name.so $objs -ldl
echo $V/libcipkkt_$name.so
