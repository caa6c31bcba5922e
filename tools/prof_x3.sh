#!/bin/bash
# rocprofv3 kernel stats of tools/time_x3.py (fp32-MFMA and split-product iterations of the headline shape, or the shapes given)
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/x3prof
mkdir -p $OUT
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_x3 -o run --output-format csv -- \
    python3 $GRAFT_REPO_ROOT/tools/time_x3.py "$@" > $OUT/run.log 2>&1)
cp /tmp/prof_x3/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find /tmp/prof_x3 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
cat $OUT/run.log | grep "ms per"
head -8 $OUT/kernel_stats.csv | cut -c1-150
