#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2quick
mkdir -p $OUT
timeout -k 10 600 python -m pytest "$@" -q -x > $OUT/pytest.out 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $OUT/pytest.out
