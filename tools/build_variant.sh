#!/bin/bash
# tools/build_variant.sh <name> <file.hip> [-DMACRO=...]: _ab/lib_<name>.so = the shipped objects with <file.hip> recompiled under the given
# macros (timing / scheduling experiments, A/B on one box with tools/ab_libs.sh). Run `make -C focnerf_amd/csrc` first.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
cd "$R/focnerf_amd/csrc"
obj=/tmp/foc_variant_${name}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-pass-failed "$@" -c "$src" -o "$obj"
objs=""
for o in raymarching gridencoder gridencoder_nd freqencoder ffmlp ffmlp_wide field_fwd combine fixedstep head densitygrid occrender occtrain; do
  if [ "$o.hip" = "$src" ]; then objs="$objs $obj"; else objs="$objs _obj/$o.o"; fi
done
mkdir -p "$R/_ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/_ab/lib_${name}.so" $objs
echo "built $R/_ab/lib_${name}.so"
