#!/bin/bash
# A/B of PREBUILT engine libraries on the GPU box (cross-compiled in the container, they travel with the snapshot):
#   tools/ab_prebuilt.sh "<bench args>" name1 name2 ...   -> photobundle_amd/libpba_hip_<name>.so ("main" = libpba_hip.so), twice each
ARGS="$1"; shift
for rep in 1 2; do
  for nm in "$@"; do
    lib=photobundle_amd/libpba_hip_$nm.so; [ "$nm" = main ] && lib=photobundle_amd/libpba_hip.so
    PBA_LIB=$lib bash tools/ab_bench.sh "V=$nm" $ARGS
  done
done
