#!/bin/bash
OUT=gpurun_out/${1:-dev}; mkdir -p $OUT
[ -f photobundle_amd/libpba_hip_timing.so ] || make -s -C photobundle_amd/csrc TIMING=1 OUT=../libpba_hip_timing.so > $OUT/make_timing.log 2>&1   # (normally cross-compiled in the container: it travels with the snapshot)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/rcp_probe.hip -o /tmp/rcp_probe 2>/dev/null && /tmp/rcp_probe | tee $OUT/rcp_probe.txt
for f in 8 16; do PBA_LIB=photobundle_amd/libpba_hip_timing.so PBA_ASYNC=0 PBA_SCHUR_TIMING=3 python tools/solve_phase_timing.py $f 50000 2>&1 | grep -E "solve_blocked|k_reduce_solve" | tail -2 | tee -a $OUT/solve_phase.txt; done
