#!/bin/bash
# A/B of diagnostic builds (libpsp_hip_abl<V>.so, built with extra -D flags) against the shipped library
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2abl
mkdir -p $OUT
for V in 0 1 2 3; do
  L=path-space-pde-solver_amd/csrc/libpsp_hip_abl$V.so; [ $V -eq 0 ] && L=path-space-pde-solver_amd/csrc/libpsp_hip.so
  [ -f $L ] || continue
  for W in "$@"; do
    PSP_LIB_PATH=$PWD/$L timeout -k 10 300 python3 bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/${W}_abl$V.json 2> $OUT/${W}_abl$V.err
    python3 -c "
import json
d=json.loads([l for l in open('$OUT/${W}_abl$V.json') if l.startswith('{')][-1]); print('variant=$V $W', 'fwd %.3f bwd %.3f ms'%(d['roofline']['fwd_kernel_ms'], d['roofline']['bwd_kernel_ms']))"
  done
done
