#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2all
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest_all.out 2> $OUT/pytest_all.err; rc=$?
echo "pytest rc=$rc"; tail -5 $OUT/pytest_all.out
[ $rc -eq 124 ] && exit 124
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.out 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.out
