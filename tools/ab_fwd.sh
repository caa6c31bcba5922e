#!/bin/bash
# A/B on ONE box: tools/time_fwd_store.py against two builds of the library (PSP_LIB_PATH), alternating
set -o pipefail
OUT=gpurun_out/ab_fwd; mkdir -p $OUT
A=${1:-path-space-pde-solver_amd/csrc/libpsp_hip_old.so}
B=${2:-path-space-pde-solver_amd/csrc/libpsp_hip.so}
for rep in 1 2; do
  for L in $A $B; do
    echo "== $L" | tee -a $OUT/ab.txt
    PSP_LIB_PATH=$PWD/$L timeout -k 10 200 python tools/time_fwd_store.py 2>/dev/null | tee -a $OUT/ab.txt || exit 1
  done
done
