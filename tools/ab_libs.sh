#!/bin/bash
# A/B of compile-time switches ON THE GPU BOX: tools/ab_libs.sh "<bench args>" name1 "<EXTRA defs 1>" name2 "<EXTRA defs 2>" ...
# builds photobundle_amd/libpba_hip_<name>.so per variant and prints one bench line each (same box, back to back, twice)
ARGS="$1"; shift
names=()
while [ $# -gt 0 ]; do
  make -s -C photobundle_amd/csrc EXTRA="$2" OUT=../libpba_hip_$1.so > /dev/null 2>&1 || echo "build of $1 failed"
  names+=("$1"); shift; shift
done
for rep in 1 2; do
  for nm in "${names[@]}"; do
    PBA_LIB=photobundle_amd/libpba_hip_$nm.so bash tools/ab_bench.sh "V=$nm" $ARGS
  done
done
