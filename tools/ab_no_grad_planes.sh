#!/bin/bash
# Upper bound of "the scatter fed from the sigma network's backward" (VERDICT round 2, next #4): a TIMING build (results wrong) in which the
# fused MLP backward writes no [L,B,C] gradient planes and the binned scatter reads none, against the shipped library, on ONE box.
# Build (in the container):  hipcc ... -DFOC_TIMING_NO_GRAD_PLANES -c gridencoder.hip / ffmlp.hip -> _ab/lib_nogradplanes.so   
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
REPS=${REPS:-2} bash "$R/tools/ab_libs.sh" "$R/_ab/lib_base.so" "$R/_ab/lib_nogradplanes.so"
for lib in lib_base lib_nogradplanes; do
  export FOCNERF_LIB_PATH="$R/_ab/$lib.so"
  bash "$R/tools/quick_stats.sh" "$lib" | grep -E "k_gbin_scatter|k_gbin_reduce|k_mlp_bwd_fusedILi64ELi2"
done
unset FOCNERF_LIB_PATH
