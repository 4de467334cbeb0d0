#!/bin/bash
# Dev tool: device ISA of the engine (tools/isa_dump.sh out.s [extra defs]); then tools/asm_stats.py out.s <kernel substring>
OUT=${1:-/tmp/isa/pba_engine.s}; shift
mkdir -p $(dirname $OUT)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPBA_PHASE_TIMING=0 "$@" -S --cuda-device-only -o $OUT -x hip /root/repo/photobundle_amd/csrc/pba_engine.hip 2>/dev/null
